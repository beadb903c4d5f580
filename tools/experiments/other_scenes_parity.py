"""Round 6: the reference's OTHER scenes (scratch/other/*.mcsd.gz: classroom, dining-room, box, two more matpreview materials — the general
surface / full-feature pool-walk kernels outside LDS, merged queries) on the GPU against the oracle, bit for bit, at a small film and spp.
The oracle is the checker here (oracle/checkers.py), as in tests/.    python tools/experiments/other_scenes_parity.py [w h spp]"""
import gzip, json, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import checkers
from _pkg import load_package
pkg = load_package()
checkers.build(ref=False)
oracle = checkers.Oracle()
w, h, spp = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (320, 180, 16)
tmp = tempfile.mkdtemp(dir="/tmp")
for name in ("box", "classroom", "dining-room", "matpreview-rough-plastic", "matpreview-thin-dielectric"):
    src = os.path.join(ROOT, "scratch", "other", name + ".mcsd.gz")
    if not os.path.exists(src):
        print(json.dumps({"scene": name, "skipped": "not in scratch/other"})); continue
    cfg = pkg.capi.Config.from_mcsd_bytes(gzip.open(src, "rb").read()).set_film(w, h, spp)
    path = os.path.join(tmp, name + ".mcsd")
    cfg.save_mcsd(path)
    r = pkg.capi.Renderer(cfg, device=0)
    frame, st = r.draw()
    kernel = r.last_kernel()
    r.close()
    want, info = oracle.render(path)
    differing = int((frame != want).any(axis=2).sum())
    print(json.dumps({"scene": name, "film": [w, h, spp], "pixels_differing": differing, "exact": differing == 0, "kernel_ms": round(st["kernel_milliseconds"], 2),
                      "oracle_seconds": round(info["seconds"], 1), "kernel": kernel[:130]}), flush=True)
