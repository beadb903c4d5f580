"""Independent-sample mode: explicit sample splits on rank shares (one GPU), stream kernel vs lanes kernel."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from _pkg import load_package
pkg = load_package()
rows = []
for name, film in (("dragon", (1280, 720, 256)), ("matpreview-rc", (1024, 1024, 128)), ("cornell", (512, 512, 256))):
    r = pkg.capi.Renderer(pkg.workloads.config(name, *film), device=0)
    r.draw()
    for n in (1, 2, 8):
        rng = pkg.capi.TileRange(0, n, 0)
        buf = torch.zeros(r.tiles_in(rng) * 64 * 3, dtype=torch.float32, device="cuda:0")
        for split in (0, 1, 2, 4, 8, 16, 32):
            r.set_rng(1, seed=1, sample_split=split)
            r.draw_device(buf.data_ptr(), rng, packed=True)
            best = min(r.draw_device(buf.data_ptr(), rng, packed=True)["kernel_milliseconds"] for _ in range(2))
            row = {"workload": name, "n_gpus": n, "split": split, "rank0_ms": round(best, 2), "mean": round(float(buf.mean()), 6), "kernel": r.last_kernel()[:40] + " ... " + r.last_kernel()[-28:]}
            print(json.dumps(row), flush=True)
            rows.append(row)
    r.close()
json.dump(rows, open("gpurun_out/split_share.json", "w"), indent=1)
