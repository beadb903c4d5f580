#!/usr/bin/env python3
"""cornell 512x512 spp 256 (and volumetric-caustic): pixel order of the lanes kernel x compaction of thinning workgroups.
Run once per MCPT_COMPACT setting (the switch is read when the library first draws):
    MCPT_COMPACT=1 python tools/experiments/cornell_order_compact.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from _pkg import load_package
pkg = load_package()
for name in sys.argv[1:] or ["cornell"]:
    w, h, spp = pkg.workloads.WORKLOADS[name][1]
    r = pkg.capi.Renderer(pkg.workloads.config(name), device=0)
    for order in (-1, 0, 1):
        r.set_pixel_order(order)
        r.draw()
        ms = []
        for _ in range(4):
            _, st = r.draw()
            ms.append(st["kernel_milliseconds"])
        print(json.dumps({"workload": name, "compact": os.environ.get("MCPT_COMPACT", "1"), "pixel_order": order, "kernel_ms": min(ms),
                          "msamples_per_s": w * h * spp / min(ms) / 1e3, "kernel": r.last_kernel()}), flush=True)
    r.close()
