#!/bin/bash
# round 6, GPU session 27: dragon's class with the path market and its tiles most expensive first (the library's rule now); the GPU suite; rank shares
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r06_s27; mkdir -p $O
timeout 600 python tools/experiments/market_counts.py dragon 20 > $O/market_dragon.jsonl 2> $O/market_dragon.err; cat $O/market_dragon.jsonl | cut -c1-400; tail -3 $O/market_dragon.err
timeout 1800 python -m pytest tests/ -m gpu -q > $O/gpu_tests.log 2>&1
grep -E "passed|failed|FAILED|Error" $O/gpu_tests.log | tail -8
timeout 900 python tools/ab_libraries.py --workloads dragon --share 8 --draws 6 --rounds 2 final=monte-carlo-path-tracing_amd/libmcpt_hip.so > $O/share8.jsonl 2> $O/share8.err; cut -c1-300 $O/share8.jsonl
timeout 900 python bench.py --workload dragon --steps 10 > $O/bench_dragon.json 2> $O/bench_dragon.err; tail -c 900 $O/bench_dragon.json
