#!/bin/bash
# round 6, GPU session 10: node records 144 bytes apart in LDS (pool_walk.h, kPoolLdsNodeVecs) against 128 — frame times and LDS counters
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r06_s10; mkdir -p $O
L=monte-carlo-path-tracing_amd
timeout 900 python tools/ab_libraries.py --workloads cornell,volumetric --draws 10 --rounds 2 pad144=$L/libmcpt_hip.so pad128=$L/exp/nodes128/libmcpt_hip.so > $O/ab.jsonl 2> $O/ab.err
cut -c1-400 $O/ab.jsonl
G="SQ_INSTS_LDS,SQ_ACTIVE_INST_LDS,SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE;SQ_WAVE_CYCLES,SQ_WAIT_ANY,SQ_WAIT_INST_LDS,SQ_ACTIVE_INST_VALU,SQ_INSTS_VALU,SQ_BUSY_CYCLES"
for w in cornell volumetric; do
  timeout 600 python tools/pmc_groups.py --out $O/pmc_${w}_pad144.json --groups "$G" -- python tools/render_scene.py workload:$w --draws 1 > $O/pmc_${w}_pad144.log 2>&1
  MCPT_LIB=$PWD/$L/exp/nodes128/libmcpt_hip.so timeout 600 python tools/pmc_groups.py --out $O/pmc_${w}_pad128.json --groups "$G" -- python tools/render_scene.py workload:$w --draws 1 > $O/pmc_${w}_pad128.log 2>&1
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_s10/pmc_*.json')):
    d=json.load(open(f)); c=d.get('counters',d)
    print(f, {k:c[k] for k in c if k.startswith('SQ_')})
P
