#!/usr/bin/env python3
"""Renders one scene a few times on cuda:0 and prints throughput and the walk
counters — the command tools/pmc_profile.py wraps for scenes other than the bench's.

    python tools/render_scene.py builtin:cornell-box | workload:<cornell|dragon|matpreview-rc|matpreview-rd|volumetric> | standin:blob-field | standin:terrain | file.mcsd | scene.xml
                                 [--film W H SPP] [--draws N] [--counted]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("scene")
    ap.add_argument("--film", type=int, nargs=3, default=None, metavar=("W", "H", "SPP"))
    ap.add_argument("--draws", type=int, default=2)
    ap.add_argument("--counted", action="store_true", help="also run the counting instantiation once")
    ap.add_argument("--kernel", choices=["auto", "stream", "lanes"], default="auto",
                    help="kernel formulation: the library's choice by scene class (default), the stream "
                         "kernel, or the lane-owns-a-path kernel")
    ap.add_argument("--work", type=int, default=-1, help="work distribution: -1 library's choice, 0 fixed lists, 1 work counter")
    ap.add_argument("--prepass", type=int, default=-1, help="camera-ray pre-pass: -1 library's choice, 0 off, 1 on")
    ap.add_argument("--kernel-mode", type=int, default=None, help="numeric kernel mode (overrides --kernel)")
    ap.add_argument("--tile-order", type=int, default=-1, help="hand-out order of the tiles: -1 library's choice, 0 image order, 1 most expensive first, 2 image order in XCD bands")
    a = ap.parse_args()
    from _pkg import load_package
    pkg = load_package()
    capi = pkg.capi
    if a.scene.startswith("builtin:"):
        cfg = capi.Config.builtin(a.scene[8:])
    elif a.scene == "standin:blob-field":
        cfg = capi.Config.from_scene(pkg.scenes.blob_field_scene())
    elif a.scene == "standin:terrain":
        cfg = capi.Config.from_scene(pkg.scenes.terrain_scene(640, 1280, 720, 256))
    elif a.scene.startswith("workload:"):
        cfg = pkg.workloads.config(a.scene[9:])
    elif a.scene.endswith(".xml"):
        cfg = capi.Config.load_xml(a.scene)
    else:
        cfg = capi.Config.load_mcsd(a.scene)
    if a.film:
        cfg.set_film(*a.film)
    w, h, spp = cfg.film()
    r = capi.Renderer(cfg)
    r.set_kernel({"auto": -1, "stream": 1, "lanes": 0}[a.kernel] if a.kernel_mode is None else a.kernel_mode)
    r.set_work_distribution(a.work).set_prepass(a.prepass).set_tile_order(a.tile_order)
    out = {"scene": a.scene, "film": [w, h, spp], "info": r.info()}
    for _ in range(a.draws):
        _, st = r.draw()
    out["kernel"] = r.last_kernel()
    out["kernel_ms"] = st["kernel_milliseconds"]
    out["msamples_per_s"] = w * h * spp / st["kernel_milliseconds"] / 1e3
    if a.counted:
        _, c = r.draw(counted=True)
        rays = c["closest_rays"] + c["shadow_rays"]
        out["counted"] = {
            "rays_per_sample": rays / c["samples"], "node_visits_per_ray": c["node_tests"] / (4 if "pool-walk" in r.last_kernel() else 2) / rays,
            "prim_tests_per_ray": c["prim_tests"] / rays,
            "node_phase_lane_utilisation": (c["node_tests"] / (4 if "pool-walk" in r.last_kernel() else 2)) / (64.0 * max(c["wave_node_steps"], 1)),
            "prim_phase_lane_utilisation": c["prim_tests"] / (64.0 * max(c["wave_prim_steps"], 1)),
        }
    print(json.dumps(out))


if __name__ == "__main__":
    main()
