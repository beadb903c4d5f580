import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("MCPT_CHECK_WALKS", "0")
from _pkg import load_package
pkg = load_package()
S = pkg.scenes; M = pkg.mcsd
def run(scene, name):
    r = pkg.capi.Renderer(pkg.capi.Config.from_scene(scene), device=0)
    r.set_pool_walk(0); g, _ = r.draw(); k0 = r.last_kernel(); r.set_pool_walk(-1)
    f, _ = r.draw()
    print(json.dumps({"case": name, "equal": float(1 - (f != g).any(axis=2).mean()), "kernel": r.last_kernel()[:70], "k0": k0[:40]}), flush=True)
    r.close()
sc = S.cornell_box(64, 64, 2)
sc.emitters.append(M.Emitter(type=M.EMIT_DIRECTIONAL, direction=(0.2, -0.7, -0.68), radiance=(3, 3, 3)))
run(sc, "cornell + directional (area light deferred, emitter immediate)")
sc = S.cornell_box(64, 64, 2)
sc.instances = sc.instances[:-1]   # no area light
sc.emitters.append(M.Emitter(type=M.EMIT_DIRECTIONAL, direction=(0.2, -0.7, -0.68), radiance=(3, 3, 3)))
run(sc, "cornell without its light + directional (emitter deferred)")
