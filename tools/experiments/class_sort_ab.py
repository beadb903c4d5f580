"""Class-sorted kernel against the unsorted lanes kernel on the LDS-resident full-feature workloads (same process, alternating)."""
import sys, os, json, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from _pkg import load_package
pkg = load_package()
jobs = [("volumetric", (1280, 720, int(sys.argv[1]) if len(sys.argv) > 1 else 128))]
for name, film in jobs:
    r = pkg.capi.Renderer(pkg.workloads.config(name, *film), device=0)
    for rep in range(2):
        for sort in (0, 1):
            r.set_class_sort(sort)
            r.draw()
            best = min(r.draw()[1]["kernel_milliseconds"] for _ in range(3))
            frame, _ = r.draw()
            print(json.dumps({"workload": name, "film": film, "class_sort": sort, "kernel": r.last_kernel(), "ms": round(best, 3),
                              "msamples": round(film[0] * film[1] * film[2] / best / 1e3, 1), "sha": hashlib.sha256(frame.tobytes()).hexdigest()[:12]}), flush=True)
    r.close()
