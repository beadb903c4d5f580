#!/usr/bin/env python3
"""How full is the GPU over a frame?  Renders a workload with MCPT_WAVE_CLOCK set (capi.cpp: start / end time of every wavefront
of the render launch on the constant 100 MHz clock) and prints the number of wavefronts alive over time in tenths of the frame.

    python tools/experiments/wave_timeline.py dragon [--share N] [--spread S] [--out gpurun_out/wave_timeline_dragon.json]
"""
import argparse
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload")
    ap.add_argument("--share", type=int, default=1)
    ap.add_argument("--spread", type=int, default=0)
    ap.add_argument("--pool", type=int, default=-1, help="mcpt_renderer_set_pool_walk")
    ap.add_argument("--sort", type=int, default=-1, help="mcpt_renderer_set_class_sort")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    path = os.path.join(tempfile.mkdtemp(dir="/tmp"), "clock.bin")
    os.environ["MCPT_WAVE_CLOCK"] = path
    import numpy as np
    from _pkg import load_package
    pkg = load_package()
    r = pkg.capi.Renderer(pkg.workloads.config(a.workload), device=0)
    if a.spread:
        r.set_lane_spread(a.spread)
    if a.pool >= 0:
        r.set_pool_walk(a.pool)
    if a.sort >= 0:
        r.set_class_sort(a.sort)
    if a.share > 1:
        import torch
        rng = pkg.capi.TileRange(0, a.share, 0)
        buf = torch.zeros(r.tiles_in(rng) * 64 * 3, dtype=torch.float32, device="cuda:0")
        draw = lambda: r.draw_device(buf.data_ptr(), rng, packed=True)
    else:
        draw = lambda: r.draw()[1]
    draw()
    st = draw()
    c = np.fromfile(path, dtype=np.uint64).reshape(-1, 4) if os.path.exists(path) else np.zeros((0, 4), np.uint64)
    c = c[c[:, 1] > 0].astype(np.int64)
    if len(c) == 0:   # (a kernel without the clocks: the class-sorted one)
        print(json.dumps({"workload": a.workload, "kernel": r.last_kernel(), "kernel_ms": st["kernel_milliseconds"], "kernel_ms_third_draw": draw()["kernel_milliseconds"]}))
        return
    t0, t1 = c[:, 0].min(), c[:, 1].max()
    span = float(t1 - t0)
    grid = np.linspace(0.0, 1.0, 21)
    alive = [int(((c[:, 0] - t0 <= g * span) & (c[:, 1] - t0 > g * span)).sum()) for g in grid]
    life = (c[:, 1] - c[:, 0]) / span
    rec = {"workload": a.workload, "share": a.share, "kernel": r.last_kernel(), "kernel_ms": st["kernel_milliseconds"],
           "span_ms_100MHz_clock": span / 1e5, "wavefronts": int(len(c)),
           "alive_at_twentieths_of_the_frame": alive,
           "mean_occupancy_of_the_launch": float(life.mean()),
           "wavefront_lifetime_quantiles_10_50_90_100": [float(np.quantile(life, q)) for q in (0.1, 0.5, 0.9, 1.0)],
           "end_time_quantiles_10_50_90": [float(np.quantile((c[:, 1] - t0) / span, q)) for q in (0.1, 0.5, 0.9)]}
    # the wavefronts that end last: when did they take their last pixel, how long did that last batch run, how many pixels had they
    last = np.argsort(c[:, 1])[-16:]
    rec["last_16_wavefronts"] = [{"end": round(float((c[i, 1] - t0) / span), 3), "last_pixel_taken_at": round(float((c[i, 2] - t0) / span), 3),
                                  "pixels": int(c[i, 3])} for i in last]
    took = c[c[:, 3] > 0]
    rec["last_pixel_taken_at_quantiles_50_90_100"] = [float(np.quantile((took[:, 2] - t0) / span, q)) for q in (0.5, 0.9, 1.0)]
    rec["pixels_per_wavefront_quantiles_10_50_90_100"] = [float(np.quantile(c[:, 3], q)) for q in (0.1, 0.5, 0.9, 1.0)]
    print(json.dumps(rec))
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        json.dump(rec, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
