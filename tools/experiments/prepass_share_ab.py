"""Round 6: rank 0's tile share of an N-GPU job of an LDS-resident workload with the camera-ray pre-pass off (the library's choice there) and
on — a share is a few long chains on a mostly idle GPU, and with the pre-pass a sample that ends starts its successor in the same step
(path_core.h resolve_and_regenerate).    python tools/experiments/prepass_share_ab.py cornell,volumetric"""
import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from _pkg import load_package
pkg = load_package()
for name in (sys.argv[1] if len(sys.argv) > 1 else "cornell").split(","):
    r = pkg.capi.Renderer(pkg.workloads.config(name), device=0)
    for n in (1, 2, 4, 8):
        rng = pkg.capi.TileRange(0, n, 0)
        buf = torch.zeros(r.tiles_in(rng) * 64 * 3, dtype=torch.float32, device="cuda:0")
        rec = {"workload": name, "n_gpus": n}
        sha = {}
        for mode in (0, 1, 0, 1):
            r.set_prepass(mode)
            r.draw_device(buf.data_ptr(), rng, packed=True)
            ms = sorted(r.draw_device(buf.data_ptr(), rng, packed=True)["kernel_milliseconds"] for _ in range(4))
            key = "prepass" if mode else "no_prepass"
            rec[key] = round(min(rec.get(key, 1e9), ms[len(ms) // 2]), 2)
            sha[mode] = hashlib.sha256(buf.cpu().numpy().tobytes()).hexdigest()[:12]
            rec["kernel_" + key] = r.last_kernel()[:100]
        rec["frames_identical"] = sha[0] == sha[1]
        print(json.dumps(rec), flush=True)
    r.close()
