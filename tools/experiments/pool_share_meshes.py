"""Rank shares of the mesh workloads with the pool-walk kernels at explicit lanes-per-path (RenderJob::lane_spread): which spread
is best for a share of N = 2, 4, 8 GPUs?  (auto = the launcher's rule.)"""
import sys, os, json, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from _pkg import load_package
pkg = load_package()
FILMS = {"dragon": (1280, 720, 256), "matpreview-rc": (1024, 1024, 128), "matpreview-rd": (1024, 1024, 128)}
rows = []
for name in (sys.argv[1:] or ["dragon", "matpreview-rc"]):
    for n in (1, 2, 4, 8):
        digests = set()
        for spread in (0, 1, 2, 4, 8):
            if n == 1 and spread > 1:
                continue
            r = pkg.capi.Renderer(pkg.workloads.config(name, *FILMS[name]), device=0)
            r.set_lane_spread(spread)
            rng = pkg.capi.TileRange(0, n, 0)
            buf = torch.zeros(r.tiles_in(rng) * 64 * 3, dtype=torch.float32, device="cuda:0")
            r.draw_device(buf.data_ptr(), rng, packed=True)
            best = min(r.draw_device(buf.data_ptr(), rng, packed=True)["kernel_milliseconds"] for _ in range(3))
            digests.add(hashlib.sha256(buf.cpu().numpy().tobytes()).hexdigest()[:12])
            row = {"workload": name, "n_gpus": n, "lane_spread": spread or "auto", "rank0_ms": round(best, 2), "kernel": r.last_kernel()[:80]}
            print(json.dumps(row), flush=True)
            rows.append(row)
            r.close()
        print(json.dumps({"n_gpus": n, "same_frames": len(digests) == 1}), flush=True)
json.dump(rows, open("gpurun_out/pool_share_meshes.json", "w"), indent=1)
