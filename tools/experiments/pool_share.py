"""Pool walk against one walk per lane on rank 0's tile share of an N-GPU run (N = 1, 2, 4, 8), one GPU: the time of the
share (the strong-scaling bound), frames compared.  usage: pool_share.py [workload ...]"""
import sys, os, json, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from _pkg import load_package
pkg = load_package()
FILMS = {"cornell": (512, 512, 256), "volumetric": (1280, 720, 256)}
rows = []
for name in (sys.argv[1:] or ["cornell"]):
    film = FILMS[name]
    for n in (1, 2, 4, 8):
        digests, base = {}, None
        for label, pool, spread in (("lane", 0, 0), ("lane-spread1", 0, 1), ("lane-spread2", 0, 2), ("lane-spread4", 0, 4), ("pool", 1, 0), ("pool-spread1", 1, 1), ("pool-spread2", 1, 2), ("pool-spread4", 1, 4), ("pool-spread8", 1, 8)):
            if n == 1 and spread > 1:
                continue
            r = pkg.capi.Renderer(pkg.workloads.config(name, *film), device=0)
            r.set_pool_walk(pool).set_lane_spread(spread)
            rng = pkg.capi.TileRange(0, n, 0)
            buf = torch.zeros(r.tiles_in(rng) * 64 * 3, dtype=torch.float32, device="cuda:0")
            r.draw_device(buf.data_ptr(), rng, packed=True)
            best = min(r.draw_device(buf.data_ptr(), rng, packed=True)["kernel_milliseconds"] for _ in range(3))
            digests[label] = hashlib.sha256(buf.cpu().numpy().tobytes()).hexdigest()[:12]
            base = base or best
            row = {"workload": name, "film": film, "n_gpus": n, "config": label, "rank0_ms": round(best, 3), "vs_lane": round(base / best, 3),
                   "kernel": r.last_kernel()[:100]}
            print(json.dumps(row), flush=True)
            rows.append(row)
            r.close()
        print(json.dumps({"n_gpus": n, "same_frames": len(set(digests.values())) == 1, "digests": digests}), flush=True)
json.dump(rows, open("gpurun_out/pool_share.json", "w"), indent=1)
