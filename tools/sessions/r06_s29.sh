#!/bin/bash
# round 6, GPU session 29: cornell — the path market in the LDS kernel (one path per ticket) against the events that deal out / pack
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r06_s29; mkdir -p $O
L=monte-carlo-path-tracing_amd
timeout 900 python tools/ab_libraries.py --workloads cornell --draws 8 --rounds 2 pack=$L/libmcpt_hip.so deal=$L/exp/deals1/libmcpt_hip.so deal_and_market=$L/exp/ldsmarket/libmcpt_hip.so > $O/ab.jsonl 2> $O/ab.err
cut -c1-200 $O/ab.jsonl; tail -2 $O/ab.err
MCPT_LIB=$PWD/$L/exp/ldsmarket/libmcpt_hip.so timeout 300 python tools/experiments/market_counts.py cornell 8 > $O/market_cornell.jsonl 2>> $O/ab.err; cat $O/market_cornell.jsonl
timeout 600 python tools/ab_libraries.py --workloads cornell --share 8 --draws 6 --rounds 2 pack=$L/libmcpt_hip.so deal=$L/exp/deals1/libmcpt_hip.so deal_and_market=$L/exp/ldsmarket/libmcpt_hip.so > $O/ab_share8.jsonl 2>> $O/ab.err; cut -c1-200 $O/ab_share8.jsonl
