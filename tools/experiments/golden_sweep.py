#!/usr/bin/env python3
"""Every golden frame through the library's default kernel (ordered walk forced): exact fraction and kernel name per case."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("MCPT_CHECK_WALKS", "0")
from _pkg import load_package
pkg = load_package()
from golden_cases import cases
sc = cases(pkg.scenes)
for name, scene in sc.items():
    want = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))["frame"]
    r = pkg.capi.Renderer(pkg.capi.Config.from_scene(scene), device=0)
    f, _ = r.draw()
    k = r.last_kernel()
    r.set_pool_walk(0)
    g, _ = r.draw()
    r.close()
    print(json.dumps({"case": name, "exact": float((f == want).all(axis=2).mean()), "exact_without_pool_walk": float((g == want).all(axis=2).mean()), "kernel": k[:70]}), flush=True)
