"""Round 6 (EXPERIMENTS R6-2): cornell's rank-0 tile share of an N-GPU run on ONE GPU with the lean LDS-resident kernel in its two
forms — two queries per vertex (mcpt_renderer_set_pool_walk 1) and merged queries (2) — and explicit lanes per path."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from _pkg import load_package
pkg = load_package()
rows = []
r = pkg.capi.Renderer(pkg.workloads.config("cornell", 512, 512, 256), device=0)
for n in (1, 2, 4, 8):
    rng = pkg.capi.TileRange(0, n, 0)
    buf = torch.zeros(r.tiles_in(rng) * 64 * 3, dtype=torch.float32, device="cuda:0")
    frames = {}
    for mode in (1, 2):
        for spread in ((0,) if n == 1 else (0, 2, 4, 8, 16)):
            r.set_pool_walk(mode).set_lane_spread(spread)
            r.draw_device(buf.data_ptr(), rng, packed=True)
            ms = sorted(r.draw_device(buf.data_ptr(), rng, packed=True)["kernel_milliseconds"] for _ in range(5))
            frames[(mode, spread)] = buf.cpu().numpy().copy()
            row = {"n_gpus": n, "pool_walk_mode": mode, "lane_spread": spread, "rank0_ms_median": ms[2], "rank0_ms_min": ms[0], "kernel": r.last_kernel()}
            print(json.dumps(row), flush=True)
            rows.append(row)
    ref = frames[(1, 0)]
    assert all((f == ref).all() for f in frames.values()), "the share's frame depends on the form"
r.close()
json.dump(rows, open("gpurun_out/r06_lean_merge_share.json", "w"), indent=1)
