import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("MCPT_CHECK_WALKS", "0")
from _pkg import load_package
pkg = load_package()
S = pkg.scenes
def run(scene, name):
    r = pkg.capi.Renderer(pkg.capi.Config.from_scene(scene), device=0)
    r.set_pool_walk(0); g, _ = r.draw(); r.set_pool_walk(-1)
    for pre in (-1, 0, 1):
        r.set_prepass(pre)
        f, _ = r.draw()
        print(json.dumps({"case": name, "prepass": pre, "equal": float(1 - (f != g).any(axis=2).mean()), "kernel": r.last_kernel()[:90]}), flush=True)
    r.close()
run(S.cornell_box(64, 64, 2), "cornell")
sc = S.cornell_box(64, 64, 2)
sp = S.uv_sphere_mesh(24, 48, 0.3, (0.0, 0.6, 0.0))
sc.instances.append(pkg.mcsd.Instance(type=pkg.mcsd.INST_MESHES, id_bsdf=2, to_world=pkg.mcsd.IDENTITY.copy(), positions=sp["positions"], normals=sp["normals"], texcoords=sp["texcoords"], indices=sp["indices"]))
run(sc, "cornell+sphere")
