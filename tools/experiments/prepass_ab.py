"""Round 6: the camera-ray pre-pass on / off (mcpt_renderer_set_prepass) now that a sample whose path ends starts its successor in the same
step (path_core.h regenerate_in_step) — LDS-resident scenes, where the pre-pass used to lose 4-5 %.  Same process, alternating."""
import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from _pkg import load_package
pkg = load_package()
for name in (sys.argv[1] if len(sys.argv) > 1 else "cornell").split(","):
    r = pkg.capi.Renderer(pkg.workloads.config(name), device=0)
    w, h, spp = pkg.workloads.WORKLOADS[name][1]
    ms, wall, sha, kernel = {0: [], 1: []}, {0: [], 1: []}, {}, {}
    for rnd in range(2):
        for mode in (0, 1):
            r.set_prepass(mode)
            r.draw()
            for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 6):
                frame, st = r.draw()
                ms[mode].append(st["kernel_milliseconds"]), wall[mode].append(st["render_seconds"] * 1e3)
            sha[mode], kernel[mode] = hashlib.sha256(frame.tobytes()).hexdigest()[:12], r.last_kernel()[:120]
    rec = {"workload": name, "frames_identical": sha[0] == sha[1]}
    for mode, label in ((0, "no_prepass"), (1, "prepass")):
        v, u = sorted(ms[mode]), sorted(wall[mode])
        rec[label] = {"kernel_median_ms": round(v[len(v) // 2], 2), "draw_wall_median_ms": round(u[len(u) // 2], 2), "msamples_by_wall": round(w * h * spp / u[len(u) // 2] / 1e3, 1), "kernel": kernel[mode]}
    print(json.dumps(rec), flush=True)
    r.close()
