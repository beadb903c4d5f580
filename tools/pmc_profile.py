#!/usr/bin/env python3
"""Collects hardware counters for the render kernel with rocprofv3, one pass per
counter group (the groups cannot share a pass), and writes one JSON summary.

    python tools/pmc_profile.py --out gpurun_out/pmc.json [--spp 32] [-- extra bench.py args]

Runs `python bench.py --spp N --steps 1 --warmup 0 --no-cpu-baseline` under
`rocprofv3 --kernel-trace --pmc <group>` and sums each counter over the dispatches
whose kernel name contains "render_kernel" / "stream_kernel" (not the counting instantiations).  Only --kernel-trace is combined with
--pmc (see the MI355X guide's profiling section)."""
import argparse
import csv
import glob
import json
import os
import subprocess
import sys

GROUPS = [   # SQ has 8 slots per pass, TCC 4 (FETCH_SIZE takes 3, WRITE_SIZE 2), GRBM 2
    ["SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU",
     "SQ_ACTIVE_INST_ANY", "SQ_INSTS_SALU"],
    ["SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_BRANCH", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_LDS_BANK_CONFLICT",
     "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS"],
    ["SQ_INSTS_SMEM", "SQ_INSTS_VALU_TRANS_F32", "SQ_INST_CYCLES_VMEM_RD", "SQ_ACTIVE_INST_SCA", "GRBM_GUI_ACTIVE",
     "TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum"],
    ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_EA0_RDREQ_sum"],
    ["FETCH_SIZE"],
    ["WRITE_SIZE"],
    ["SQC_ICACHE_REQ", "SQC_ICACHE_HITS", "SQC_ICACHE_MISSES", "SQC_ICACHE_MISSES_DUPLICATE", "SQ_IFETCH", "SQ_IFETCH_LEVEL"],
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--spp", type=int, default=32)
    ap.add_argument("--workdir", default="gpurun_out/pmc_passes")
    ap.add_argument("--command", default=None,
                    help="profile this command instead of bench.py (e.g. 'python tools/render_scene.py standin:blob-field')")
    ap.add_argument("--groups", default=None, help="comma-separated indices of the counter groups to run (default: all)")
    ap.add_argument("rest", nargs="*")
    a = ap.parse_args()
    env = dict(os.environ, TMPDIR="/tmp")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    counters, kernels, failed = {}, set(), []
    wanted = None if a.groups is None else {int(x) for x in a.groups.split(",")}
    for gi, group in enumerate(GROUPS):
        if wanted is not None and gi not in wanted:
            continue
        d = os.path.join(a.workdir, f"pass{gi}")
        os.makedirs(d, exist_ok=True)
        target = (a.command.split() if a.command else
                  [sys.executable, os.path.join(root, "bench.py"), "--spp", str(a.spp), "--steps", "1", "--warmup", "0",
                   "--no-cpu-baseline", *a.rest])
        cmd = ["rocprofv3", "--kernel-trace", "--pmc", *group, "--output-format", "csv", "-d", d, "-o", "p", "--", *target]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if r.returncode != 0 or not files:
            failed.append({"group": group, "rc": r.returncode, "stderr": r.stderr[-400:]})
            continue
        for row in csv.DictReader(open(files[0])):
            # the plain render kernel only: bench.py also launches the counting
            # instantiation (second template argument true) for its bookkeeping
            name = row["Kernel_Name"]
            if ("render_kernel" not in name and "stream_kernel" not in name and "sorted_kernel" not in name) or \
                    (", true, " in name and "render_kernel" in name) or ("u, true, " in name and "stream_kernel" in name):
                continue
            kernels.add(row["Kernel_Name"][:120])
            counters[row["Counter_Name"]] = counters.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    c = counters
    derived = {}
    if c.get("SQ_ACTIVE_INST_VALU") and c.get("SQ_THREAD_CYCLES_VALU"):
        # thread-cycles / (instruction-cycles * 64 lanes)
        derived["valu_lane_utilization"] = c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64.0)
    if c.get("SQ_WAVE_CYCLES"):
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_LDS",
                  "SQ_ACTIVE_INST_LDS"):
            if k in c:
                derived[k.lower() + "_per_wave_cycle"] = c[k] / c["SQ_WAVE_CYCLES"]
    if c.get("SQ_INSTS_VALU"):
        for k in ("SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_BRANCH"):
            if k in c:
                derived[k.lower() + "_per_valu"] = c[k] / c["SQ_INSTS_VALU"]
    if c.get("SQC_ICACHE_REQ"):
        derived["icache_miss_rate"] = c.get("SQC_ICACHE_MISSES", 0.0) / c["SQC_ICACHE_REQ"]
        if c.get("SQ_IFETCH"):
            derived["ifetch_level_per_fetch"] = c.get("SQ_IFETCH_LEVEL", 0.0) / c["SQ_IFETCH"]
    if "FETCH_SIZE" in c:
        derived["hbm_read_bytes_note"] = "FETCH_SIZE/WRITE_SIZE are in KiB; apply the guide's gfx950 correction"
        derived["fetch_kib"] = c["FETCH_SIZE"]
        derived["write_kib"] = c.get("WRITE_SIZE")
    out = {"command": "rocprofv3 --kernel-trace --pmc <group> -- %s (one pass per group; counters summed over "
                      "the dispatches of the plain render kernel)" %
                      (a.command or "python bench.py --spp %d --steps 1 --warmup 0 --no-cpu-baseline %s"
                       % (a.spp, " ".join(a.rest))),
           "kernels": sorted(kernels), "counters": counters, "derived": derived, "failed_groups": failed}
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    json.dump(out, open(a.out, "w"), indent=1)
    print(json.dumps({"derived": derived, "failed": [f["group"] for f in failed]}))


if __name__ == "__main__":
    main()
