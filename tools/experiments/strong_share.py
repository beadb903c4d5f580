"""Strong-scaling bound measured on ONE GPU: the time of rank 0's tile share of an N-GPU run (N = 1, 2, 4, 8), for the
reference random stream and for the independent-sample mode (samples of a pixel split over lanes)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from _pkg import load_package
pkg = load_package()
rows = []
for name, film in (("cornell", (512, 512, 256)), ("dragon", (1280, 720, 256)), ("matpreview-rc", (1024, 1024, 128)), ("volumetric", (1280, 720, 256))):
    for rng_mode in (0, 1):
        r = pkg.capi.Renderer(pkg.workloads.config(name, *film), device=0)
        if rng_mode:
            r.set_rng(1, seed=1, sample_split=0)
        base = None
        for n in (1, 2, 4, 8):
            rng = pkg.capi.TileRange(0, n, 0)
            buf = torch.zeros(r.tiles_in(rng) * 64 * 3, dtype=torch.float32, device="cuda:0")
            r.draw_device(buf.data_ptr(), rng, packed=True)
            best = min(r.draw_device(buf.data_ptr(), rng, packed=True)["kernel_milliseconds"] for _ in range(3))
            base = base or best
            row = {"workload": name, "film": film, "rng": "pcg" if rng_mode else "reference", "n_gpus": n, "rank0_ms": best,
                   "speedup_bound": base / best, "kernel": r.last_kernel()}
            print(json.dumps(row), flush=True)
            rows.append(row)
        r.close()
json.dump(rows, open("gpurun_out/strong_share.json", "w"), indent=1)
