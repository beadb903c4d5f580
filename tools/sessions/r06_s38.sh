#!/bin/bash
# round 6, GPU session 38: rounds of in-step regeneration (1 / 2 / 4 = production / 8 / 16) on dragon and matpreview; the pre-pass on the LDS-resident workloads with it
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r06_s38; mkdir -p $O
L=monte-carlo-path-tracing_amd
timeout 1800 python tools/ab_libraries.py --workloads dragon,matpreview-rc,matpreview-rd --draws 5 --rounds 2 k1=$L/exp/regen1/libmcpt_hip.so k2=$L/exp/regen2/libmcpt_hip.so k4=$L/libmcpt_hip.so k8=$L/exp/regen8/libmcpt_hip.so k16=$L/exp/regen16/libmcpt_hip.so > $O/ab_rounds.jsonl 2> $O/ab.err
cut -c1-200 $O/ab_rounds.jsonl; tail -2 $O/ab.err
timeout 900 python tools/experiments/prepass_ab.py cornell,volumetric 6 > $O/prepass_lds.jsonl 2>> $O/ab.err; cat $O/prepass_lds.jsonl | cut -c1-600
