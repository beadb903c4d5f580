import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("MCPT_CHECK_WALKS", "0")
from _pkg import load_package
pkg = load_package()
S = pkg.scenes
for (w, spp) in ((64, 1), (512, 1)):
    scene = S.cornell_box(w, w, spp)
    r = pkg.capi.Renderer(pkg.capi.Config.from_scene(scene), device=0)
    r.set_pool_walk(0); g, _ = r.draw()
    r.set_pool_walk(-1)
    for spread in (0, 1, 2, 64):
        for order in (-1, 0):
            r.set_lane_spread(spread).set_pixel_order(order)
            f, _ = r.draw()
            d = (f != g).any(axis=2)
            bad = np.argwhere(d)
            print(json.dumps({"w": w, "spread": spread, "order": order, "equal": float(1 - d.mean()), "kernel": r.last_kernel()[:70],
                              "bad_rows_hist": np.bincount(bad[:, 0] * 8 // w, minlength=8).tolist() if len(bad) else []}), flush=True)
    r.close()
