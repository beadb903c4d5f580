#!/usr/bin/env python3
"""Closest-hit rate of the lean trace-only kernels on incoherent secondary rays (mcpt_debug_trace_rate).

    python tools/experiments/trace_rate.py [--workloads dragon,matpreview-rd] [--rays 2000000]

Rays: the camera's first hits on the workload's film become origins, directions are drawn cosine-weighted about the
surface normal, the batch is shuffled (what a diffuse bounce feeds a trace launch).  Prints one JSON line per
(workload, kernel): G rays/s, and whether every kernel names the same primitive for every ray.
  full-4 / full-8   one ray per lane, full traversal stack per lane in LDS (walk_depth entries), compiled for 4 / 8 wavefronts per SIMD
  short-R           one ray per lane, short stack: R entries per lane in LDS, older entries in HBM (short_stack.h), 8 per SIMD
  wide-4 / wide-8   one ray per lane on the 4-wide quantised hierarchy (64-byte nodes with four children), short stack"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", default="dragon,matpreview-rd")
    ap.add_argument("--rays", type=int, default=2_000_000)  # (mode 3 takes at most 2^21)
    a = ap.parse_args()
    from _pkg import load_package
    pkg = load_package()
    rng = np.random.default_rng(11)
    for name in a.workloads.split(","):
        w, h, _ = pkg.workloads.WORKLOADS[name][1]
        cfg = pkg.workloads.config(name, w, h, 1)
        r = pkg.capi.Renderer(cfg, device=0)
        # camera rays through random film positions (the renderer's own camera: table "camera" = eye, front, dx, dy)
        c = r.table("camera")
        eye, front, dx, dy = c[0:3], c[3:6], c[6:9], c[9:12]
        n0 = a.rays * 3
        x = rng.uniform(-1, 1, n0).astype(np.float32)[:, None]
        y = rng.uniform(-1, 1, n0).astype(np.float32)[:, None]
        d = front[None, :] + x * dx[None, :] + y * dy[None, :]
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        o = np.tile(eye[None, :], (n0, 1)).astype(np.float32)
        out, _ = r.debug_intersect(o, d.astype(np.float32))
        hit = out[:, 0] > 0
        pos, nrm = out[hit, 7:10], out[hit, 10:13]
        k = min(len(pos), a.rays)
        pos, nrm = pos[:k], nrm[:k]
        # cosine-weighted directions about the normal
        u1, u2 = rng.uniform(0, 1, k), rng.uniform(0, 1, k)
        rr, phi = np.sqrt(u1), 2 * np.pi * u2
        t = np.where(np.abs(nrm[:, :1]) > 0.9, np.array([[0, 1, 0]]), np.array([[1, 0, 0]]))
        b1 = np.cross(nrm, t)
        b1 /= np.linalg.norm(b1, axis=1, keepdims=True)
        b2 = np.cross(nrm, b1)
        dirs = (rr * np.cos(phi))[:, None] * b1 + (rr * np.sin(phi))[:, None] * b2 + np.sqrt(1 - u1)[:, None] * nrm
        dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
        rays = np.concatenate([pos + 1e-4 * nrm, dirs], axis=1).astype(np.float32)
        rays = rays[rng.permutation(k)]
        ref = None
        for label, mode, waves, ring in (("full-4", 0, 4, 0), ("full-8", 0, 8, 0), ("short-4", 2, 8, 4), ("short-8", 2, 8, 8),
                                         ("short-16", 2, 8, 16), ("wide-4", 3, 4, 0), ("wide-8", 3, 8, 0)):
            best = None
            for _ in range(3):
                found, ms = r.trace_rate(rays, mode, waves, ring)
                best = ms if best is None else min(best, ms)
            if ref is None:
                ref = found
            print(json.dumps({"workload": name, "kernel": label, "rays": k, "ms": best, "grays_per_s": k / best / 1e6,
                              "hit_fraction": float((found != 0xFFFFFFFF).mean()), "same_answers": bool(np.array_equal(found, ref)),
                              "walk_depth": r.info()["walk_depth"]}), flush=True)
        r.close()


if __name__ == "__main__":
    main()
