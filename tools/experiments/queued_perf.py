#!/usr/bin/env python3
"""Queued renderer (set_kernel mode 5) against the calibrated single-kernel choice on the mesh workloads.

    python tools/experiments/queued_perf.py [--spp-div D] [--workloads dragon,matpreview-rc,...] [--pools 0,64,128]

Prints one JSON line per (workload, configuration): milliseconds of the second draw, Msamples/s, the kernel string,
and whether the frame hash equals the reference configuration's."""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spp-div", type=int, default=4)
    ap.add_argument("--workloads", default="dragon,matpreview-rc,matpreview-rd")
    ap.add_argument("--pools", default="0,64,128")
    ap.add_argument("--draws", type=int, default=2)
    ap.add_argument("--tile-orders", default="", help="e.g. 0,1: every configuration with image-order and cost-ordered tile hand-out")
    ap.add_argument("--modes", default="", help="also these set_kernel modes (0 lanes, 1 stream, 4 stream in wavefront rounds)")
    a = ap.parse_args()
    from _pkg import load_package
    pkg = load_package()
    for name in a.workloads.split(","):
        w, h, spp = pkg.workloads.WORKLOADS[name][1]
        spp = max(spp // a.spp_div, 1)
        cfg = pkg.workloads.config(name, w, h, spp)
        r = pkg.capi.Renderer(cfg, device=0)
        ref_hash = None
        configs = [("auto", -1, 0)] + [(f"queued pool {p}", 5, int(p)) for p in a.pools.split(",") if p != ""]
        configs += [({0: "lanes", 1: "stream", 4: "stream, wavefront rounds"}[int(m)], int(m), 0) for m in a.modes.split(",") if m != ""]
        if a.tile_orders:
            configs = [(f"{label}, tile order {o}", mode, pool, int(o)) for (label, mode, pool) in configs for o in a.tile_orders.split(",")]
        else:
            configs = [(label, mode, pool, -1) for (label, mode, pool) in configs]
        for label, mode, pool, order in configs:
            r.set_kernel(mode, slots=pool)
            r.set_tile_order(order)
            t0 = time.perf_counter()
            frame, st = r.draw()
            first = time.perf_counter() - t0
            ms = []
            for _ in range(a.draws):
                t0 = time.perf_counter()
                frame, st = r.draw()
                ms.append(1e3 * (time.perf_counter() - t0))
            digest = hashlib.sha256(frame.tobytes()).hexdigest()
            if ref_hash is None:
                ref_hash = digest
            print(json.dumps({"workload": name, "film": [w, h, spp], "config": label, "first_draw_ms": 1e3 * first,
                              "draw_ms": min(ms), "kernel_ms": st["kernel_milliseconds"],
                              "msamples_per_s": w * h * spp / min(ms) / 1e3, "same_frame": digest == ref_hash,
                              "kernel": r.last_kernel()[:160]}), flush=True)
        r.close()


if __name__ == "__main__":
    main()
