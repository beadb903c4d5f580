"""MCPT_COST_ORDER experiment on cornell: each setting in its own process (the switch is read once), alternating."""
import sys, os, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, os, json, hashlib
sys.path.insert(0, %r)
from _pkg import load_package
pkg = load_package()
name, w, h, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
r = pkg.capi.Renderer(pkg.workloads.config(name, w, h, spp), device=0)
_, first = r.draw()
ms = sorted(r.draw()[1]["kernel_milliseconds"] for _ in range(5))
frame, st = r.draw()
print(json.dumps({"workload": name, "film": [w, h, spp], "cost_order": os.environ.get("MCPT_COST_ORDER", "0") + " " + os.environ.get("MCPT_COMPACT", "") + " " + os.environ.get("MCPT_COST_PROBE_SPP", "") + " L" + os.environ.get("MCPT_COST_LAYOUT", "") + " r1=" + os.environ.get("MCPT_COST_R1", ""), "kernel": r.last_kernel(), "first_ms": round(first["kernel_milliseconds"], 3),
                  "ms_best": round(ms[0], 3), "ms_median": round(ms[2], 3), "msamples": round(w * h * spp / ms[0] / 1e3, 1), "sha": hashlib.sha256(frame.tobytes()).hexdigest()[:12]}))
''' % ROOT
jobs = [j.split(":") for j in (sys.argv[1] if len(sys.argv) > 1 else "cornell:512:512:256").split(",")]
for name, w, h, spp in jobs:
    for rep in range(2):
        for mode in (sys.argv[2].split(",") if len(sys.argv) > 2 else ("0", "1", "2")):
            extra = dict(kv.split("=") for kv in mode.split("+")[1:])
            env = dict(os.environ, MCPT_COST_ORDER=mode.split("+")[0], **extra)
            p = subprocess.run([sys.executable, "-c", CHILD, name, w, h, spp], env=env, capture_output=True, text=True)
            print(p.stdout.strip().splitlines()[-1] if p.stdout.strip() else json.dumps({"mode": mode, "error": p.stderr[-500:]}), flush=True)
