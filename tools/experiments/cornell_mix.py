#!/usr/bin/env python3
"""cornell 512 x 512 spp 256: one tile per wavefront laid out by probed cost (the library's choice) against the transposed pixel
order (a wavefront holds one pixel of 64 tiles and thins out) — under the pool walk a thinned wavefront's paths run faster."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from _pkg import load_package
pkg = load_package()
import hashlib
out = {}
for name, setup in (("library", lambda r: r), ("transposed", lambda r: r.set_pixel_order(1)), ("tiles_image_order", lambda r: r.set_pixel_order(0).set_tile_order(0))):
    r = pkg.capi.Renderer(pkg.workloads.config("cornell"), device=0)
    setup(r)
    ms = []
    for _ in range(6):
        f, st = r.draw()
        ms.append(st["kernel_milliseconds"])
    out[name] = {"ms": ms[1:], "kernel": r.last_kernel(), "sha": hashlib.sha256(f.tobytes()).hexdigest()[:12]}
    r.close()
print(json.dumps(out, indent=1))
