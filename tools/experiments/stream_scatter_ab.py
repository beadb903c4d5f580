"""Transposed hand-out in the stream kernel (set_pixel_order 1) against tile order, mesh workloads, same process."""
import sys, os, json, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from _pkg import load_package
pkg = load_package()
jobs = [("dragon", (1280, 720, 128)), ("matpreview-rc", (1024, 1024, 64)), ("matpreview-rd", (1024, 1024, 64))]
for name, film in jobs:
    r = pkg.capi.Renderer(pkg.workloads.config(name, *film), device=0)
    for rep in range(2):
        for order in (-1, 1):
            r.set_pixel_order(order)
            r.draw()
            best = min(r.draw()[1]["kernel_milliseconds"] for _ in range(3))
            frame, _ = r.draw()
            print(json.dumps({"workload": name, "film": film, "pixel_order": order, "kernel": r.last_kernel()[:90], "ms": round(best, 3),
                              "msamples": round(film[0] * film[1] * film[2] / best / 1e3, 1), "sha": hashlib.sha256(frame.tobytes()).hexdigest()[:12]}), flush=True)
    r.close()
