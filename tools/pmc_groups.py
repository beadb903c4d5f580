#!/usr/bin/env python3
"""Hardware counters of the render kernels for ONE command, one rocprofv3 pass per counter group (only --kernel-trace is
combined with --pmc).  The memory-side groups of round 5 (TCP / TCC hit rates, pending-stall cycles, TA busy) are the default.

    python tools/pmc_groups.py --out gpurun_out/pmc_dragon.json [--groups "A,B;C,D"] -- python tools/render_scene.py workload:dragon --draws 1

Counters are summed over the dispatches whose kernel name contains one of KERNEL_WORDS and is not a counting instantiation;
the summary carries the derived rates (TCP hit rate = 1 - TCC read requests / cache accesses, L2 hit rate = TCC_HIT / (HIT + MISS),
bytes the L2 asked the fabric for = TCC_EA0_RDREQ x 64 B (32 B requests counted apart where the box has the counter))."""
import argparse
import csv
import glob
import json
import os
import subprocess
import sys
import tempfile

DEFAULT_GROUPS = [
    ["TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum", "TCP_PENDING_STALL_CYCLES_sum", "TCP_TOTAL_ACCESSES_sum"],
    ["TCP_GATE_EN1_sum", "TCP_GATE_EN2_sum", "TCP_TCP_TA_DATA_STALL_CYCLES_sum", "TCP_TA_TCP_STATE_READ_sum"],
    ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_EA0_RDREQ_sum", "TCC_REQ_sum"],
    ["TCC_EA0_RDREQ_32B_sum", "TCC_READ_sum", "TCC_EA0_RD_UNCACHED_32B_sum", "TCC_TAG_STALL_sum"],
    ["TA_BUSY_avr", "TA_BUSY_max", "TA_TA_BUSY_sum", "TA_ADDR_STALLED_BY_TC_CYCLES_sum", "TA_DATA_STALLED_BY_TC_CYCLES_sum"],
    ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_INSTS_VMEM_RD",
     "SQ_BUSY_CYCLES"],
    ["SQ_INST_CYCLES_VMEM_RD", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_LDS", "SQ_INSTS_LDS", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_SALU",
     "GRBM_GUI_ACTIVE"],
]
KERNEL_WORDS = ("render_kernel", "sorted_kernel", "stream_kernel", "queued_")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--groups", default=None, help="'A,B;C,D': counter groups, one pass each (default: the memory-side set)")
    ap.add_argument("--timeout", type=int, default=600)
    ap.add_argument("command", nargs=argparse.REMAINDER)
    a = ap.parse_args()
    cmd_tail = a.command[1:] if a.command and a.command[0] == "--" else a.command
    groups = DEFAULT_GROUPS if a.groups is None else [g.split(",") for g in a.groups.split(";") if g]
    env = dict(os.environ, TMPDIR="/tmp")
    counters, kernels, failed, durations = {}, set(), [], {}
    # a pass fails as a whole when it names a counter this rocprofv3 does not know: ask for the list first and drop those
    dropped = []
    try:
        listing = subprocess.run(["rocprofv3", "-L"], env=env, capture_output=True, text=True, timeout=120)
        known = listing.stdout + listing.stderr
        if len(known) > 1000:
            dropped = [n for g in groups for n in g if n not in known]
            groups = [[n for n in g if n in known] for g in groups]
            groups = [g for g in groups if g]
    except Exception as e:  # noqa: BLE001 (a box without the listing: try the groups as they are)
        dropped = [f"listing failed: {e}"]
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        for gi, group in enumerate(groups):
            d = os.path.join(tmp, f"pass{gi}")
            cmd = ["rocprofv3", "--kernel-trace", "--pmc", *group, "--output-format", "csv", "-d", d, "-o", "p", "--", *cmd_tail]
            try:
                r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=a.timeout)
            except subprocess.TimeoutExpired:
                failed.append({"group": group, "error": "timeout"})
                continue
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                # a group with a counter this box does not know fails as a whole: retry its counters one by one
                failed.append({"group": group, "rc": r.returncode, "stderr": r.stderr[-300:]})
                continue
            for row in csv.DictReader(open(files[0])):
                name = row["Kernel_Name"]
                if not any(k in name for k in KERNEL_WORDS) or ", true, " in name:
                    continue
                kernels.add(name.split("(")[0][-90:])
                counters[row["Counter_Name"]] = counters.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
            for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
                ns = 0
                for row in csv.DictReader(open(f)):
                    if any(k in row["Kernel_Name"] for k in KERNEL_WORDS) and ", true, " not in row["Kernel_Name"]:
                        ns += int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
                durations[gi] = ns * 1e-6
    c = counters
    derived = {}
    if c.get("TCP_TOTAL_CACHE_ACCESSES_sum"):
        derived["tcp_hit_rate"] = 1.0 - c.get("TCP_TCC_READ_REQ_sum", 0.0) / c["TCP_TOTAL_CACHE_ACCESSES_sum"]
    if c.get("TCC_HIT_sum") is not None and (c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0)) > 0:
        derived["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    if c.get("TCC_EA0_RDREQ_sum"):
        rd32 = c.get("TCC_EA0_RDREQ_32B_sum", 0.0)
        derived["l2_fabric_read_bytes"] = (c["TCC_EA0_RDREQ_sum"] - rd32) * 64.0 + rd32 * 32.0
    if c.get("SQ_WAVE_CYCLES"):
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS"):
            if k in c:
                derived[k.lower() + "_per_wave_cycle"] = c[k] / c["SQ_WAVE_CYCLES"]
    if c.get("TCP_GATE_EN1_sum") and c.get("TCP_PENDING_STALL_CYCLES_sum") is not None:
        derived["tcp_pending_stall_per_gate_en1"] = c["TCP_PENDING_STALL_CYCLES_sum"] / c["TCP_GATE_EN1_sum"]
    out = {"command": "rocprofv3 --kernel-trace --pmc <group> -- " + " ".join(cmd_tail), "kernels": sorted(kernels), "counters": counters,
           "derived": derived, "kernel_ms_by_pass": durations, "failed_groups": failed,
           "unknown_counters_dropped": dropped}
    os.makedirs(os.path.dirname(os.path.abspath(a.out)) or ".", exist_ok=True)
    json.dump(out, open(a.out, "w"), indent=1)
    print(json.dumps({"derived": derived, "kernel_ms_by_pass": durations, "failed": [f["group"] for f in failed]}))


if __name__ == "__main__":
    main()
