#!/usr/bin/env python3
"""Debug of the merged queries: cornell goldens, and a cornell box with a mesh sphere inside (an area light on the pool walk
OUTSIDE LDS) against the same renderer with the pool walk off."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("MCPT_CHECK_WALKS", "0")
from _pkg import load_package
pkg = load_package()
S = pkg.scenes
def both(scene, name):
    r = pkg.capi.Renderer(pkg.capi.Config.from_scene(scene), device=0)
    f, _ = r.draw(); k = r.last_kernel()
    r.set_pool_walk(0); g, _ = r.draw(); k0 = r.last_kernel(); r.close()
    d = (f != g).any(axis=2)
    print(json.dumps({"case": name, "equal_to_no_pool": float(1 - d.mean()), "kernel": k[:60], "kernel0": k0[:40], "first_bad": [int(x) for x in np.argwhere(d)[:3].ravel()]}), flush=True)
for spp in (1, 2, 8):
    both(S.cornell_box(64, 64, spp), f"cornell_64_spp{spp}")
sc = S.cornell_box(64, 64, 4)
sp = S.uv_sphere_mesh(24, 48, 0.3, (0.0, 0.6, 0.0))
sc.instances.append(pkg.mcsd.Instance(type=pkg.mcsd.INST_MESHES, id_bsdf=2, to_world=pkg.mcsd.IDENTITY.copy(), positions=sp["positions"], normals=sp["normals"], texcoords=sp["texcoords"], indices=sp["indices"]))
both(sc, "cornell_with_mesh_sphere_spp4")
