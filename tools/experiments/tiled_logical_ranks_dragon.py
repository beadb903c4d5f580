import ctypes, json, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
from _pkg import load_package
pkg = load_package()
cfg = pkg.workloads.config("dragon", 1280, 720, 8)
r = pkg.capi.Renderer(cfg, device=0)
plain, _ = r.draw()
r.close()
out = {}
for n in (2, 3, 8):
    t = pkg.capi.TiledRenderer(cfg, devices=(0,) * n, flags=pkg.capi.TILED_LOGICAL_RANKS)
    frame, st = t.draw()
    again, _ = t.draw()
    t.close()
    out[n] = [bool(np.array_equal(frame, plain)), bool(np.array_equal(again, plain))]
print(json.dumps(out))
