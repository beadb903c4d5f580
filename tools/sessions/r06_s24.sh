#!/bin/bash
# round 6, GPU session 24: the path market (paths given to the wavefronts of finished workgroups) — dragon's frame time and counters, then the GPU suite
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r06_s24; mkdir -p $O
timeout 600 python tools/experiments/market_counts.py dragon 16 > $O/market_dragon.jsonl 2> $O/market_dragon.err; cat $O/market_dragon.jsonl; tail -3 $O/market_dragon.err
timeout 1800 python -m pytest tests/ -m gpu -q > $O/gpu_tests.log 2>&1
grep -E "passed|failed|FAILED|Error" $O/gpu_tests.log | tail -8
timeout 600 python tools/experiments/market_counts.py dragon 16 >> $O/market_dragon.jsonl 2>> $O/market_dragon.err; tail -1 $O/market_dragon.jsonl
