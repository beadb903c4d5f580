import os, sys, subprocess, numpy as np, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
code = r'''
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
from _pkg import load_package
pkg = load_package()
w, h, spp = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
r = pkg.capi.Renderer(pkg.capi.Config.builtin("cornell-box").set_film(w, h, spp), device=0)
import os
if os.environ.get("DBG_SPREAD"): r.set_lane_spread(int(os.environ["DBG_SPREAD"]))
if os.environ.get("DBG_WORK"): r.set_work_distribution(int(os.environ["DBG_WORK"]))
if os.environ.get("DBG_ORDER"): r.set_pixel_order(int(os.environ["DBG_ORDER"]))
f, _ = r.draw(); print(r.last_kernel(), file=sys.stderr)
np.save(sys.argv[5], f)
'''
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for (w, h, spp, extra) in ((128, 128, 8, {}), (128, 128, 8, {"DBG_SPREAD": "1"}), (128, 128, 8, {"DBG_WORK": "0"}), (128, 128, 8, {"DBG_ORDER": "0"}), (120, 128, 8, {}), (128, 96, 8, {})):
    fr = {}
    for c in ("0", "1"):
        out = f"/tmp/f_{c}.npy"
        p = subprocess.run([sys.executable, "-c", code, root, str(w), str(h), str(spp), out], env=dict(os.environ, MCPT_COMPACT=c, **extra), capture_output=True, text=True, timeout=120)
        if p.returncode: print(p.stderr[-500:])
        fr[c] = np.load(out)
    bad = (fr["0"] != fr["1"]).any(axis=2)
    ys, xs = np.nonzero(bad)
    print((w, h, spp), extra, "px00", fr["0"][0,0].tolist(), fr["1"][0,0].tolist(), "differing pixels", int(bad.sum()), "nan", int(np.isnan(fr["1"]).sum()), "first", list(zip(ys[:8].tolist(), xs[:8].tolist())),
          "zero pixels in compact frame", int((fr["1"].sum(axis=2) == 0).sum()), "vs", int((fr["0"].sum(axis=2) == 0).sum()), p.stderr.strip()[-80:])
