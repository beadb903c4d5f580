#!/usr/bin/env python3
"""A/B of BUILDS of libmcpt_hip.so on one box (compiler flags, build parameters): every (library, workload) is a process
of its own (capi.py binds the library named by MCPT_LIB), the libraries take turns `--rounds` times, medians of the draws'
kernel times, frames compared by hash across libraries.

    python tools/ab_libraries.py [--workloads cornell,dragon,other:classroom,...] [--draws 3] [--rounds 2] name=path.so name=path.so ...
"""
import argparse
import hashlib
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(workload, draws):
    from _pkg import load_package
    pkg = load_package()
    if workload.startswith("other:"):  # the reference's other scenes (tools/convert_other_scenes.py -> scratch/other), at tools/rule_vs_calibrated.py's film
        import gzip
        cfg = pkg.capi.Config.from_mcsd_bytes(gzip.open(os.path.join(ROOT, "scratch", "other", workload[6:] + ".mcsd.gz"), "rb").read()).set_film(640, 360, 64)
    else:
        cfg = pkg.workloads.config(workload)
    r = pkg.capi.Renderer(cfg, device=0)
    if os.environ.get("MCPT_AB_RNG"):  # throughput modes: 1 PCG-hashed streams, 2 Sobol points (--rng)
        r.set_rng(int(os.environ["MCPT_AB_RNG"]))
    ms, frame = [], None
    share = int(os.environ.get("MCPT_AB_SHARE", "1"))  # > 1: rank 0's packed tile share of an N-GPU job (a near-empty GPU)
    if share > 1:
        import numpy as np
        import torch
        rng = pkg.capi.TileRange(0, share, 0)
        buf = torch.zeros(r.tiles_in(rng) * 64 * 3, dtype=torch.float32, device="cuda:0")
        for _ in range(draws + 1):
            ms.append(r.draw_device(buf.data_ptr(), rng, packed=True)["kernel_milliseconds"])
        frame = buf.cpu().numpy()
    for _ in range(draws + 1 if share <= 1 else 0):
        frame, st = r.draw()
        ms.append(st["kernel_milliseconds"])
    print(json.dumps({"ms": ms[1:], "sha": hashlib.sha256(frame.tobytes()).hexdigest()[:16], "kernel": r.last_kernel()}))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        return child(sys.argv[2], int(sys.argv[3]))
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", default="cornell,dragon,matpreview-rc,matpreview-rd,volumetric")
    ap.add_argument("--draws", type=int, default=3)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--rng", type=int, default=0, help="mcpt_renderer_set_rng mode of every renderer (0: the reference's stream)")
    ap.add_argument("--share", type=int, default=1, help="N > 1: time rank 0's tile share of an N-GPU job instead of the whole frame")
    ap.add_argument("libs", nargs="+")
    a = ap.parse_args()
    if a.rng:
        os.environ["MCPT_AB_RNG"] = str(a.rng)
    if a.share > 1:
        os.environ["MCPT_AB_SHARE"] = str(a.share)
    # name=path.so[@VAR=value[@VAR=value ...]]: environment of that arm's processes (commit-time switches)
    libs, arm_env = [], {}
    for l in a.libs:
        n, rest = l.split("=", 1)
        path, *envs = rest.split("@")
        libs.append([n, path])
        arm_env[n] = dict(e.split("=", 1) for e in envs)
    out = {}
    for w in a.workloads.split(","):
        rec = {n: {"ms": [], "sha": set()} for n, _ in libs}
        for _ in range(a.rounds):
            for n, path in libs:
                env = dict(os.environ, MCPT_LIB=os.path.abspath(path), **arm_env[n])
                p = subprocess.run([sys.executable, __file__, "--child", w, str(a.draws)], env=env, capture_output=True, text=True)
                if p.returncode != 0:
                    rec[n]["error"] = p.stderr[-400:]
                    continue
                d = json.loads(p.stdout.strip().split("\n")[-1])
                rec[n]["ms"] += d["ms"]
                rec[n]["sha"].add(d["sha"])
                rec[n]["kernel"] = d["kernel"]
        shas = set().union(*(rec[n]["sha"] for n in rec))
        out[w] = {"frames_identical": len(shas) == 1,
                  **{n: {"median_ms": statistics.median(rec[n]["ms"]) if rec[n]["ms"] else None, "min_ms": min(rec[n]["ms"], default=None), "max_ms": max(rec[n]["ms"], default=None), "all_ms": [round(v, 2) for v in rec[n]["ms"]],
                         "n": len(rec[n]["ms"]), "kernel": rec[n].get("kernel"), **({"error": rec[n]["error"]} if "error" in rec[n] else {})}
                     for n in rec}}
        print(json.dumps({w: out[w]}), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "ab_libraries.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
