"""Runs mcpt_renderer_check_walks (ordered walk vs the reference-order walk, bit for bit) on every BASELINE
configuration at its FULL film (spp reduced where the reference-order walk of a large mesh would take minutes)
and on the reference's other shipped scenes when their converted fixtures are present (scratch/real).

    python tests/full_size_walk_check.py [out.json]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from _pkg import load_package  # noqa: E402

pkg = load_package()
FILMS = {"cornell": (512, 512, 256), "dragon": (1280, 720, 16), "matpreview-rc": (1024, 1024, 32),
         "matpreview-rd": (1024, 1024, 32), "volumetric": (1280, 720, 256)}
rows = []
for name, film in FILMS.items():
    r = pkg.capi.Renderer(pkg.workloads.config(name, *film), device=0)
    t0 = time.time()
    n, first, worst = r.check_walks()
    rows.append({"scene": name, "film": film, "pixels_differing": n, "first_pixel": first, "max_abs_diff": worst,
                 "seconds": round(time.time() - t0, 2), "primitives": r.info()["primitives"]})
    print(json.dumps(rows[-1]), flush=True)
    r.close()
real = os.path.join(ROOT, "scratch", "real")
for f in ("classroom_constant_light", "dining_room_constant_light", "matpreview_rough_plastic", "matpreview_thin_dielectric",
          "volumetric_caustic_hg"):
    path = os.path.join(real, f + ".mcsd")
    if not os.path.exists(path):
        continue
    film = (640, 360, 8) if "room" in f else (512, 512, 16)
    r = pkg.capi.Renderer(pkg.capi.Config.load_mcsd(path).set_film(*film), device=0)
    t0 = time.time()
    n, first, worst = r.check_walks()
    rows.append({"scene": f, "film": film, "pixels_differing": n, "first_pixel": first, "max_abs_diff": worst,
                 "seconds": round(time.time() - t0, 2), "primitives": r.info()["primitives"]})
    print(json.dumps(rows[-1]), flush=True)
    r.close()
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "walk_self_check.json")
os.makedirs(os.path.dirname(out), exist_ok=True)
json.dump(rows, open(out, "w"), indent=1)
