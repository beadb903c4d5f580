#!/usr/bin/env python3
"""Where does a wavefront of the class-sorted kernel spend its time?  Renders a workload with a library whose sorted_kernel unit was
built with -DMCPT_PHASE_CLOCK=1 (tools/experiments/build_exp2.sh phase "-DMCPT_PHASE_CLOCK=1" sorted_kernel) and MCPT_WAVE_CLOCK
set, and reads the per-phase sums the kernel leaves behind the per-wavefront clocks (path_core.h, phase_mark): shader-clock
cycles, number of marks, and lanes that arrived at the marks.

    MCPT_LIB=monte-carlo-path-tracing_amd/exp/phase/libmcpt_hip.so python tools/experiments/phase_clock.py volumetric [--spp N] [--out f.json]

share          = the phase's cycles / all booked cycles (a wavefront's wall time, waiting included)
lanes_at_mark  = mean number of lanes (of 64) that reach the END of the phase: an upper bound of the lane utilisation inside it
"""
import argparse
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
PHASES = ["regenerate", "extend (closest-hit walk)", "resolve: surface frame", "resolve: medium on the segment (free-flight sampling)",
          "resolve: escape / light / back face / roulette", "class sort (barriers included)", "connect: emitter sampling", "connect: shadow walks",
          "connect: transmittance, BSDF / phase value, MIS", "connect: area-light sampling", "scatter: phase function", "scatter: BSDF",
          "scatter: rest", "resolve: surface frame of a triangle", "resolve: surface frame of a quadric"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload")
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--sort", type=int, default=-1)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    path = os.path.join(tempfile.mkdtemp(dir="/tmp"), "clock.bin")
    os.environ["MCPT_WAVE_CLOCK"] = path
    import numpy as np
    from _pkg import load_package
    pkg = load_package()
    w, h, spp = pkg.workloads.WORKLOADS[a.workload][1]
    cfg = pkg.workloads.config(a.workload, w, h, a.spp or spp)
    r = pkg.capi.Renderer(cfg, device=0)
    if a.sort >= 0:
        r.set_class_sort(a.sort)
    r.draw()
    _, st = r.draw()
    words = np.fromfile(path, dtype=np.uint64)
    tail = words[-64:].astype(np.float64)  # (RenderJob::phase_sums: the last 64 words of the file)
    n = len(PHASES)
    cycles, visits, lanes = tail[1:1 + n], tail[1 + n:1 + 2 * n], tail[1 + 2 * n:1 + 3 * n]
    total = cycles.sum()
    rec = {"workload": a.workload, "film": [w, h, a.spp or spp], "kernel": r.last_kernel(), "kernel_ms": st["kernel_milliseconds"],
           "library": os.environ.get("MCPT_LIB", "default"), "phases": []}
    for k, name in enumerate(PHASES):
        if visits[k] == 0:
            continue
        rec["phases"].append({"phase": name, "share": round(cycles[k] / total, 4), "marks": int(visits[k]),
                              "cycles_per_mark": round(cycles[k] / visits[k], 1), "lanes_at_mark": round(lanes[k] / visits[k], 2)})
    # useful lane fraction if every instruction of a phase ran at the lanes that reach its end
    rec["lane_bound"] = round(sum(p["share"] * p["lanes_at_mark"] / 64.0 for p in rec["phases"]), 4)
    text = json.dumps(rec, indent=1)
    print(text)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)) or ".", exist_ok=True)
        open(a.out, "w").write(text)


if __name__ == "__main__":
    main()
