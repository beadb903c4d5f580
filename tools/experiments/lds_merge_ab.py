#!/usr/bin/env python3
"""Round 6, EXPERIMENTS R6-1: the LDS-resident pool-walk kernel with merged queries (kPM) on cornell, library builds side by side
on one box — does the frame equal the golden one, and if not, which way is it off.
    python tools/experiments/lds_merge_ab.py prod ldsmerge ldsmerge_p0 ldsmerge_pf ldsmerge_sync ldsmerge_O1
(libraries from tools/experiments/build_exp2.sh / bisect_lean_merge.sh, selected with MCPT_LIB; one process per library;
MCPT_AB_QUICK=1: the 64 x 64 golden only)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, os, json, hashlib
import numpy as np
sys.path.insert(0, %r)
os.environ.setdefault("MCPT_CHECK_WALKS", "0")
from _pkg import load_package
pkg = load_package()
golden = np.load(os.path.join(%r, "tests", "golden", "cornell_64_spp8.npz"))["frame"]
out = {"lib": os.environ.get("MCPT_LIB", "prod").split("/")[-2] if "MCPT_LIB" in os.environ else "prod"}
def run(tag, scene, ref, **kw):
    r = pkg.capi.Renderer(pkg.capi.Config.from_scene(scene), device=0)
    if "spread" in kw:
        r.set_lane_spread(kw["spread"])
    f, st = r.draw()
    k = r.last_kernel()
    if ref is None:
        r.set_pool_walk(0); ref, _ = r.draw(); r.set_pool_walk(-1)
    r.close()
    d = (f != ref).any(axis=2)
    out[tag] = {"kernel": k[:70], "equal": round(float(1 - d.mean()), 4), "darker": round(float((f.sum(2) < ref.sum(2)).mean()), 4),
                "brighter": round(float((f.sum(2) > ref.sum(2)).mean()), 4), "sha": hashlib.sha256(f.tobytes()).hexdigest()[:10], "ms": round(st["kernel_milliseconds"], 3)}
S = pkg.scenes
run("cornell_64_spp8", S.cornell_box(64, 64, 8), golden)
if not os.environ.get("MCPT_AB_QUICK"):
    run("cornell_64_spp8_spread64", S.cornell_box(64, 64, 8), golden, spread=64)
    sc = S.cornell_box(64, 64, 1); sc.integrator.depth_max = 1
    run("cornell_64_spp1_depth1", sc, None)
    run("cornell_512_spp64", S.cornell_box(512, 512, 64), None)
print(json.dumps(out))
''' % (ROOT, ROOT)
for lib in sys.argv[1:]:
    env = dict(os.environ)
    env.pop("MCPT_LIB", None)
    if lib != "prod":
        env["MCPT_LIB"] = os.path.join(ROOT, "monte-carlo-path-tracing_amd", "exp", lib, "libmcpt_hip.so")
    p = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    print(p.stdout.strip().splitlines()[-1] if p.stdout.strip() else json.dumps({"lib": lib, "error": p.stderr[-600:]}), flush=True)
