"""Round 6 (EXPERIMENTS R6-5): dragon/scene.xml — the layout of the quantised 4-wide hierarchy (breadth-first / treelets of N records:
MCPT_TREELET in an experiment build with the measurement hooks) x the hand-out of the tiles (image order / image order in eight XCD
bands).  One process per layout (the layout is decided at commit), hand-outs alternating inside it; median / min / max of the draws.
    MCPT_LIB=monte-carlo-path-tracing_amd/exp/hooks/libmcpt_hip.so python tools/experiments/dragon_layout_ab.py [workload] [draws]"""
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
for rep in range(2):
    for order in (0, 2):
        r.set_tile_order(order)
        r.draw()
        ms = sorted(r.draw()[1]["kernel_milliseconds"] for _ in range(draws))
        frame, _ = r.draw()
        print(json.dumps({"workload": name, "treelet": int(os.environ.get("MCPT_TREELET", "0")), "tile_order": order, "median_ms": round(ms[len(ms) // 2], 2), "min_ms": round(ms[0], 2),
                          "max_ms": round(ms[-1], 2), "msamples_median": round(w * h * spp / ms[len(ms) // 2] / 1e3, 1), "kernel": r.last_kernel()[:90],
                          "sha": hashlib.sha256(frame.tobytes()).hexdigest()[:12]}), flush=True)
''' % ROOT
name = sys.argv[1] if len(sys.argv) > 1 else "dragon"
draws = sys.argv[2] if len(sys.argv) > 2 else "10"
for treelet in (0, 8, 16, 32, 64, 128):
    env = dict(os.environ, MCPT_TREELET=str(treelet))
    p = subprocess.run([sys.executable, "-c", CHILD, name, draws], env=env, capture_output=True, text=True)
    print(p.stdout.strip() if p.stdout.strip() else json.dumps({"treelet": treelet, "error": p.stderr[-500:]}), flush=True)
