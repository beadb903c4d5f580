"""Round 6: frame times of a workload and what its last draw's path market saw (tickets taken by waiting wavefronts, records given, items finished).
    python tools/experiments/market_counts.py dragon [draws]"""
import json, os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from _pkg import load_package
pkg = load_package()
for name in (sys.argv[1] if len(sys.argv) > 1 else "dragon").split(","):
    draws = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    r = pkg.capi.Renderer(pkg.workloads.config(name), device=0)
    if os.environ.get("MCPT_SPREAD"):  # lanes per path of the launch (mcpt_renderer_set_lane_spread)
        r.set_lane_spread(int(os.environ["MCPT_SPREAD"]))
    if os.environ.get("MCPT_TILE_ORDER"):  # mcpt_renderer_set_tile_order: 2 = image order in XCD bands
        r.set_tile_order(int(os.environ["MCPT_TILE_ORDER"]))
    w, h, spp = pkg.workloads.WORKLOADS[name][1]
    r.draw()
    ms = []
    for _ in range(draws):
        frame, st = r.draw()
        ms.append(st["kernel_milliseconds"])
    v = sorted(ms)
    print(json.dumps({"workload": name, "median_ms": round(v[len(v) // 2], 2), "min_ms": round(v[0], 2), "max_ms": round(v[-1], 2), "msamples_median": round(w * h * spp / v[len(v) // 2] / 1e3, 1),
                      "market_tickets_given_finished": [int(x) for x in r.table("market")], "items": ((w + 7) // 8) * ((h + 7) // 8) * 64,
                      "spread": os.environ.get("MCPT_SPREAD", "rule"), "tile_order": os.environ.get("MCPT_TILE_ORDER", "rule"), "sha": hashlib.sha256(frame.tobytes()).hexdigest()[:12], "kernel": r.last_kernel()[:120]}), flush=True)
    r.close()
