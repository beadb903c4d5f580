import sys, os, json, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from _pkg import load_package
pkg = load_package()
for name, film in (("cornell", (512, 512, 256)), ("volumetric", (1280, 720, 128))):
    r = pkg.capi.Renderer(pkg.workloads.config(name, *film), device=0)
    for n in (8, 4, 2):
        rng = pkg.capi.TileRange(0, n, 0)
        buf = torch.zeros(r.tiles_in(rng) * 64 * 3, dtype=torch.float32, device="cuda:0")
        for mode, spread in ((0, 1), (0, 2), (4, 1), (4, 2), (4, 4), (4, 8), (4, 16), (1, 8)):
            r.set_kernel(mode).set_lane_spread(spread).set_work_distribution(1)
            r.draw_device(buf.data_ptr(), rng, packed=True)
            best = min(r.draw_device(buf.data_ptr(), rng, packed=True)["kernel_milliseconds"] for _ in range(3))
            print(json.dumps({"workload": name, "n_gpus": n, "mode": mode, "spread": spread, "ms": round(best, 2),
                              "sha": hashlib.sha256(buf.cpu().numpy().tobytes()).hexdigest()[:10], "kernel": r.last_kernel()[:50]}), flush=True)
    r.close()
