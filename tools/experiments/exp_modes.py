"""python scratch/exp_modes.py  — production library, explicit (kernel mode, refill) pairs on the mesh workloads"""
import sys, os, json, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from _pkg import load_package
pkg = load_package()
jobs = [("dragon", (1280, 720, 64)), ("matpreview-rc", (1024, 1024, 32)), ("matpreview-rd", (1024, 1024, 32))]
for name in ("classroom_constant_light", "dining_room_constant_light"):
    if os.path.exists(os.path.join("scratch", "real", name + ".mcsd")):
        jobs.append(("file:" + name, (1280, 720, 8)))
pairs = [tuple(map(int, p.split(":"))) for p in (sys.argv[1] if len(sys.argv) > 1 else "1:0,4:0,4:8,4:16,1:8").split(",")]
for name, film in jobs:
    cfg = (pkg.capi.Config.load_mcsd(os.path.join("scratch", "real", name[5:] + ".mcsd")).set_film(*film) if name.startswith("file:")
           else pkg.workloads.config(name, *film))
    r = pkg.capi.Renderer(cfg, device=0)
    for mode, refill in pairs:
        r.set_kernel(mode, 0, refill).set_work_distribution(1).set_prepass(1)
        r.draw()
        best = min(r.draw()[1]["kernel_milliseconds"] for _ in range(3))
        frame, _ = r.draw()
        print(json.dumps({"workload": name, "mode": mode, "refill": refill, "ms": round(best, 3),
                          "msamples": round(film[0] * film[1] * film[2] / best / 1e3, 1), "sha": hashlib.sha256(frame.tobytes()).hexdigest()[:12]}), flush=True)
    r.close()
