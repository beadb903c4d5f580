import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("MCPT_CHECK_WALKS", "0")
from _pkg import load_package
pkg = load_package()
scene = pkg.scenes.cornell_box(64, 64, 1)
scene.integrator.depth_max = 1
r = pkg.capi.Renderer(pkg.capi.Config.from_scene(scene), device=0)
r.set_pool_walk(0); g, _ = r.draw(); r.set_pool_walk(-1)
r.set_lane_spread(64)
f, _ = r.draw()
d = (f != g).any(axis=2)
print(r.last_kernel(), "equal", float(1 - d.mean()), "darker", float((f.sum(2) < g.sum(2)).mean()), "brighter", float((f.sum(2) > g.sum(2)).mean()))
r.close()
