"""Round 6 (EXPERIMENTS R6-10): which pixels a wavefront's lanes take from the work counter — the 32 / 64 neighbouring pixels of a tile
(mcpt_renderer_set_pixel_order 0, the choice outside LDS so far) or pixels of 64 different tiles (1, transposed) — on the kernels
outside LDS.  Same process, the orders alternate, `draws` draws each and round; frames compared by hash.
    python tools/experiments/pixel_order_ab.py dragon,matpreview-rc [draws]"""
import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from _pkg import load_package
pkg = load_package()
names = (sys.argv[1] if len(sys.argv) > 1 else "dragon").split(",")
draws = int(sys.argv[2]) if len(sys.argv) > 2 else 10
for name in names:
    r = pkg.capi.Renderer(pkg.workloads.config(name), device=0)
    w, h, spp = pkg.workloads.WORKLOADS[name][1]
    ms, sha, kernel = {0: [], 1: []}, {}, {}
    for rnd in range(2):
        for order in (0, 1):
            r.set_pixel_order(order)
            r.draw()
            for _ in range(draws):
                frame, st = r.draw()
                ms[order].append(st["kernel_milliseconds"])
            sha[order], kernel[order] = hashlib.sha256(frame.tobytes()).hexdigest()[:12], r.last_kernel()[:150]
    rec = {"workload": name, "frames_identical": sha[0] == sha[1]}
    for order, label in ((0, "tiles"), (1, "transposed")):
        v = sorted(ms[order])
        rec[label] = {"median_ms": round(v[len(v) // 2], 2), "min_ms": round(v[0], 2), "max_ms": round(v[-1], 2), "msamples_median": round(w * h * spp / v[len(v) // 2] / 1e3, 1), "kernel": kernel[order]}
    print(json.dumps(rec), flush=True)
    r.close()
