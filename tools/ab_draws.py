#!/usr/bin/env python3
"""A/B of renderer settings on one box: alternating draws of the same workload, medians, frames compared bit for bit.

    python tools/ab_draws.py <workload> [--film W H SPP] [--reps N] -- setting=value[,setting=value...] ...

settings: pool (mcpt_renderer_set_pool_walk), sort (class sort), kernel, work, prepass, order (pixel order), tile (tile order),
waves (stream waves).  Every configuration is a renderer of its own; one JSON line."""
import argparse
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SETTERS = {"pool": "set_pool_walk", "sort": "set_class_sort", "kernel": "set_kernel", "work": "set_work_distribution",
           "prepass": "set_prepass", "order": "set_pixel_order", "tile": "set_tile_order", "waves": "set_stream_waves"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload")
    ap.add_argument("--film", type=int, nargs=3, default=None)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("configs", nargs="+", help="setting=value[,setting=value...]; 'default' = nothing set")
    a = ap.parse_args()
    from _pkg import load_package
    pkg = load_package()
    film = a.film or (0, 0, 0)
    renderers = []
    for c in a.configs:
        r = pkg.capi.Renderer(pkg.workloads.config(a.workload, *film), device=0)
        if c != "default":
            for kv in c.split(","):
                k, v = kv.split("=")
                getattr(r, SETTERS[k])(int(v))
        renderers.append(r)
    ms = [[] for _ in renderers]
    digests = [set() for _ in renderers]
    kernels = [""] * len(renderers)
    for r in renderers:
        r.draw()   # first draw: probes, allocations
    for _ in range(a.reps):
        for i, r in enumerate(renderers):
            frame, st = r.draw()
            ms[i].append(st["kernel_milliseconds"])
            digests[i].add(hashlib.sha256(frame.tobytes()).hexdigest()[:16])
            kernels[i] = r.last_kernel()
    w, h, spp = renderers[0].cfg_film if hasattr(renderers[0], "cfg_film") else pkg.workloads.config(a.workload, *film).film()
    out = {"workload": a.workload, "film": [w, h, spp], "reps": a.reps, "configs": []}
    for i, c in enumerate(a.configs):
        med = float(np.median(ms[i]))
        out["configs"].append({"config": c, "median_ms": med, "min_ms": float(np.min(ms[i])), "all_ms": [round(x, 3) for x in ms[i]],
                               "msamples_per_s": w * h * spp / med / 1e3, "frame_sha": sorted(digests[i]), "kernel": kernels[i]})
    out["same_frame"] = len(set().union(*digests)) == 1
    print(json.dumps(out))
    for r in renderers:
        r.close()


if __name__ == "__main__":
    main()
