"""Rank 0's tile share of an N-GPU strong-scaling run on ONE GPU, dense lanes against spread lanes."""
import sys, os, json, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from _pkg import load_package
pkg = load_package()
rows = []
jobs = [("cornell", (512, 512, 256)), ("dragon", (1280, 720, 64)), ("volumetric", (1280, 720, 128)), ("matpreview-rc", (1024, 1024, 32))]
spreads = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,0,2,4,8,16").split(",")]
for name, film in jobs:
    r = pkg.capi.Renderer(pkg.workloads.config(name, *film), device=0)
    r.draw()   # calibration on the whole frame
    base = {}
    for n in (1, 2, 4, 8):
        rng = pkg.capi.TileRange(0, n, 0)
        buf = torch.zeros(r.tiles_in(rng) * 64 * 3, dtype=torch.float32, device="cuda:0")
        for spread in spreads:
            r.set_lane_spread(spread)
            r.draw_device(buf.data_ptr(), rng, packed=True)
            best = min(r.draw_device(buf.data_ptr(), rng, packed=True)["kernel_milliseconds"] for _ in range(3))
            sha = hashlib.sha256(buf.cpu().numpy().tobytes()).hexdigest()[:10]
            base.setdefault(spread, best)
            row = {"workload": name, "film": film, "n_gpus": n, "spread": spread, "rank0_ms": round(best, 3),
                   "speedup_bound": round(base[spread] / best, 2), "vs_dense_1gpu": round(base[spreads[0]] / best, 2), "sha": sha,
                   "kernel": r.last_kernel()[:48]}
            print(json.dumps(row), flush=True)
            rows.append(row)
    r.close()
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rows, open("gpurun_out/spread_share.json", "w"), indent=1)
