#!/bin/bash
# round 6, GPU session 28: cornell — what the compaction's events do with a thinning workgroup's paths: pack (production), deal out from event 1 / 2 / 3 on, nothing (MCPT_COMPACT=0)
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r06_s28; mkdir -p $O
L=monte-carlo-path-tracing_amd
timeout 900 python tools/ab_libraries.py --workloads cornell --draws 8 --rounds 2 pack=$L/libmcpt_hip.so deal_from_1=$L/exp/deals1/libmcpt_hip.so deal_from_2=$L/exp/deals2/libmcpt_hip.so deal_from_3=$L/exp/deals3/libmcpt_hip.so > $O/ab.jsonl 2> $O/ab.err
cut -c1-200 $O/ab.jsonl; tail -2 $O/ab.err
MCPT_COMPACT=0 timeout 300 python tools/ab_libraries.py --workloads cornell --draws 8 --rounds 1 no_events=$L/exp/hooks/libmcpt_hip.so > $O/ab_off.jsonl 2> $O/ab_off.err; cut -c1-300 $O/ab_off.jsonl
