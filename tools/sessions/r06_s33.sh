#!/bin/bash
# round 6, GPU session 33: dragon with the path market — lanes per path of the launch (1 / 2 = the rule / 4)
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r06_s33; mkdir -p $O
for s in 1 2 4 1 2; do MCPT_SPREAD=$s timeout 400 python tools/experiments/market_counts.py dragon 12 >> $O/spread.jsonl 2>> $O/err.log; tail -1 $O/spread.jsonl | cut -c1-330; done
