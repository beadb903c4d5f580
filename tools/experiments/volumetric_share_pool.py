#!/usr/bin/env python3
"""volumetric-caustic, rank 0's tile share of an N-GPU run on ONE GPU (N = 1, 2, 4, 8; reference stream, 1280 x 720 spp 256): the
class-sorted kernel with one walk per lane (the library's choice) against the same kernel with the wavefront-cooperative pool walk
(mcpt_renderer_set_pool_walk(1) selects it there) and against the unsorted lanes kernel.  Frames hashed."""
import hashlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from _pkg import load_package
pkg = load_package()
rows = []
r = pkg.capi.Renderer(pkg.workloads.config("volumetric", 1280, 720, 256), device=0)
for label, setup in (("class-sorted, per-lane walk (default)", lambda: r.set_pool_walk(-1).set_class_sort(-1)),
                     ("class-sorted, pool walk", lambda: r.set_pool_walk(1).set_class_sort(-1)),
                     ("unsorted, per-lane walk", lambda: r.set_pool_walk(0).set_class_sort(0))):
    setup()
    for n in (1, 2, 4, 8):
        rng = pkg.capi.TileRange(0, n, 0)
        buf = torch.zeros(r.tiles_in(rng) * 64 * 3, dtype=torch.float32, device="cuda:0")
        r.draw_device(buf.data_ptr(), rng, packed=True)
        best = min(r.draw_device(buf.data_ptr(), rng, packed=True)["kernel_milliseconds"] for _ in range(3))
        row = {"config": label, "n_gpus": n, "rank0_ms": round(best, 2), "kernel": r.last_kernel()[:90], "sha": hashlib.sha256(buf.cpu().numpy().tobytes()).hexdigest()[:12]}
        print(json.dumps(row), flush=True)
        rows.append(row)
r.close()
json.dump(rows, open("gpurun_out/volumetric_share_pool.json", "w"), indent=1)
