#!/bin/bash
# round 6, GPU session 36: cornell — cost layout 1 (a workgroup = four wavefronts of one cost quarter) against 4 (latin square: every workgroup holds all four quarters)
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r06_s36; mkdir -p $O
export MCPT_LIB=$PWD/monte-carlo-path-tracing_amd/exp/hooks/libmcpt_hip.so
for l in 1 4 3 1 4; do MCPT_COST_LAYOUT=$l timeout 300 python tools/ab_libraries.py --workloads cornell --draws 10 --rounds 1 layout_$l=$MCPT_LIB >> $O/ab.jsonl 2>> $O/ab.err; tail -1 $O/ab.jsonl | cut -c1-200; done
MCPT_COST_LAYOUT=4 MCPT_COMPACT=0 timeout 300 python tools/ab_libraries.py --workloads cornell --draws 10 --rounds 1 layout_4_no_events=$MCPT_LIB >> $O/ab.jsonl 2>> $O/ab.err; tail -1 $O/ab.jsonl | cut -c1-200
