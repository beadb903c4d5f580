#!/bin/bash
# round 6, GPU session 25: the path market with a back-off in the waiting wavefronts' polling
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r06_s25; mkdir -p $O
timeout 600 python tools/experiments/market_counts.py dragon 20 > $O/market_dragon.jsonl 2> $O/market_dragon.err; cat $O/market_dragon.jsonl; tail -3 $O/market_dragon.err
timeout 600 python tools/experiments/market_counts.py dragon 20 >> $O/market_dragon.jsonl 2>> $O/market_dragon.err; tail -1 $O/market_dragon.jsonl
