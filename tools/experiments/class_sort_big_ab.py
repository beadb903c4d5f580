"""Round 6 (EXPERIMENTS R6-4): the class-sorted kernel OUTSIDE LDS against the production kernels on matpreview (BASELINE config 4,
the "BSDF-sort path"): same process, alternating; also the unsorted kernel without lanes per path by tile cost (tile order 0), which is
what the sort gives up."""
import sys, os, json, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from _pkg import load_package
pkg = load_package()
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 512
for name in ("matpreview-rc", "matpreview-rd"):
    film = (1024, 1024, spp)
    r = pkg.capi.Renderer(pkg.workloads.config(name, *film), device=0)
    for rep in range(2):
        for label, sort, tile_order in (("production", -1, -1), ("class-sorted", 1, -1), ("unsorted, image order", 0, 0), ("class-sorted, image order", 1, 0)):
            r.set_class_sort(sort).set_tile_order(tile_order)
            r.draw()
            ms = sorted(r.draw()[1]["kernel_milliseconds"] for _ in range(3))
            frame, _ = r.draw()
            print(json.dumps({"workload": name, "film": film, "form": label, "kernel": r.last_kernel(), "ms_median": round(ms[1], 3), "ms_min": round(ms[0], 3),
                              "msamples": round(film[0] * film[1] * film[2] / ms[1] / 1e3, 1), "sha": hashlib.sha256(frame.tobytes()).hexdigest()[:12]}), flush=True)
    r.close()
