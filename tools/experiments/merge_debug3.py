import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("MCPT_CHECK_WALKS", "0")
from _pkg import load_package
pkg = load_package()
S = pkg.scenes
for depth in (1,):
    for rr in (5,):
        scene = S.cornell_box(64, 64, 1)
        scene.integrator.depth_max, scene.integrator.depth_rr = depth, rr
        r = pkg.capi.Renderer(pkg.capi.Config.from_scene(scene), device=0)
        r.set_pool_walk(0); g, _ = r.draw()
        r.set_pool_walk(-1); f, _ = r.draw()
        d = (f != g).any(axis=2); bad = np.argwhere(d)
        ex = [(int(y), int(x), [float(v) for v in f[y, x]], [float(v) for v in g[y, x]]) for y, x in bad[:3]]
        print(json.dumps({"depth_max": depth, "rr": rr, "equal": float(1 - d.mean()), "merged_brighter": float((f.sum(2) > g.sum(2)).mean()), "merged_darker": float((f.sum(2) < g.sum(2)).mean()), "examples": ex}), flush=True)
        r.close()
