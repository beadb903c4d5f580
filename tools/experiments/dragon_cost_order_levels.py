"""Round 6 (EXPERIMENTS R6-10): dragon/scene.xml — the work counter's hand-out in image order (the library's choice for jobs whose
camera rays mostly miss) against most-expensive-first by the 2-spp probe with lanes per path by tile cost (MCPT_COST_ORDER=4,
MCPT_LEVEL_KAPPA, MCPT_LEVELS: a library built with -DMCPT_MEASUREMENT_HOOKS=1).  One process per setting, 12 draws.
    MCPT_LIB=monte-carlo-path-tracing_amd/exp/hooks/libmcpt_hip.so python tools/experiments/dragon_cost_order_levels.py [workload] [draws]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, os, json, hashlib
sys.path.insert(0, %r)
from _pkg import load_package
pkg = load_package()
name, draws = sys.argv[1], int(sys.argv[2])
r = pkg.capi.Renderer(pkg.workloads.config(name), device=0)
w, h, spp = pkg.workloads.WORKLOADS[name][1]
r.draw()
ms = sorted(r.draw()[1]["kernel_milliseconds"] for _ in range(draws))
frame, _ = r.draw()
print(json.dumps({"workload": name, "cost_order": os.environ.get("MCPT_COST_ORDER", "default"), "levels": os.environ.get("MCPT_LEVELS", "default"),
                  "kappa": os.environ.get("MCPT_LEVEL_KAPPA", "default"), "median_ms": round(ms[len(ms) // 2], 2), "min_ms": round(ms[0], 2),
                  "max_ms": round(ms[-1], 2), "msamples_median": round(w * h * spp / ms[len(ms) // 2] / 1e3, 1), "kernel": r.last_kernel()[:140],
                  "sha": hashlib.sha256(frame.tobytes()).hexdigest()[:12]}), flush=True)
''' % ROOT
name = sys.argv[1] if len(sys.argv) > 1 else "dragon"
draws = sys.argv[2] if len(sys.argv) > 2 else "12"
settings = [{}, {"MCPT_COST_ORDER": "4", "MCPT_LEVELS": "0"}] + [{"MCPT_COST_ORDER": "4", "MCPT_LEVEL_KAPPA": k} for k in ("0.8", "0.6", "0.45", "0.3", "0.2", "0.12")] + [{}]
for s in settings:
    env = dict(os.environ, **s)
    p = subprocess.run([sys.executable, "-c", CHILD, name, draws], env=env, capture_output=True, text=True)
    print(p.stdout.strip() if p.stdout.strip() else json.dumps({"setting": s, "error": p.stderr[-500:]}), flush=True)
