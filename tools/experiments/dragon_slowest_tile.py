"""Round 6 (EXPERIMENTS R6-10): what the LAST wavefronts of dragon/scene.xml's frame are doing.  Finds the most expensive 8 x 8 tile of the
film at full spp (rows of tiles timed one launch each, then the tiles of the slowest rows) and times that tile ALONE at 1 path per
1 / 2 / 4 / 8 / 16 / 64 lanes: the chain of a pixel's 256 samples with more and more helper lanes, on an otherwise empty GPU.
    python tools/experiments/dragon_slowest_tile.py [workload]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from _pkg import load_package
pkg = load_package()
name = sys.argv[1] if len(sys.argv) > 1 else "dragon"
cfg = pkg.workloads.config(name)
w, h, spp = pkg.workloads.WORKLOADS[name][1]
r = pkg.capi.Renderer(cfg, device=0)
tx, ty = (w + 7) // 8, (h + 7) // 8
buf = torch.zeros(tx * 64 * 3 * 4, dtype=torch.float32, device="cuda:0")


def ms_of(rng, draws=2):
    best = None
    for _ in range(draws + 1):
        t = r.draw_device(buf.data_ptr(), rng, packed=True)["kernel_milliseconds"]
        best = t if best is None else min(best, t)
    return best


if os.environ.get("MCPT_TILES"):  # "x,y;x,y": skip the search, time these tiles alone
    for xy in os.environ["MCPT_TILES"].split(";"):
        x, y = (int(v) for v in xy.split(","))
        rec = {"tile": [x, y], "ms_at_lanes_per_path": {}, "library": os.environ.get("MCPT_LIB", "default")}
        for spread in (1, 2, 4, 8, 16, 64):
            r.set_lane_spread(spread)
            rec["ms_at_lanes_per_path"][spread] = round(ms_of(pkg.capi.TileRange(y * tx + x, 1, 1), 2), 2)
        print(json.dumps(rec), flush=True)
    sys.exit(0)
rows = [(ms_of(pkg.capi.TileRange(y * tx, 1, tx), 1), y) for y in range(ty)]
rows.sort(reverse=True)
print(json.dumps({"slowest_rows_ms": [(round(t, 2), y) for t, y in rows[:6]], "fastest_row_ms": round(rows[-1][0], 2), "rows": ty, "tiles_per_row": tx}), flush=True)
tiles = []
for t_row, y in rows[:3]:
    for x in range(tx):
        tiles.append((ms_of(pkg.capi.TileRange(y * tx + x, 1, 1), 0), x, y))
tiles.sort(reverse=True)
print(json.dumps({"slowest_tiles_ms_alone": [(round(t, 2), x, y) for t, x, y in tiles[:8]], "kernel": r.last_kernel()[:120]}), flush=True)
for t_tile, x, y in tiles[:3]:
    rec = {"tile": [x, y], "ms_at_lanes_per_path": {}}
    for spread in (1, 2, 4, 8, 16, 64):
        r.set_lane_spread(spread)
        rec["ms_at_lanes_per_path"][spread] = round(ms_of(pkg.capi.TileRange(y * tx + x, 1, 1), 2), 2)
    r.set_lane_spread(0)
    rec["kernel"] = r.last_kernel()[:120]
    print(json.dumps(rec), flush=True)
# the same tile among its neighbours: the 2 x 2 tiles around it, and its whole row, at the launch's own rule
t_tile, x, y = tiles[0]
print(json.dumps({"tile": [x, y], "row_ms": round(ms_of(pkg.capi.TileRange(y * tx, 1, tx), 2), 2), "whole_frame_ms": round(ms_of(pkg.capi.TileRange(0, 1, 0), 2), 2)}), flush=True)
