#!/bin/bash
# round 6, GPU session 22: the GPU suite on the library with the tail spread; dragon's frame over time with it; dragon's bench line
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r06_s22; mkdir -p $O
timeout 1500 python -m pytest tests/ -m gpu -x -q > $O/gpu_tests.log 2>&1
grep -E "passed|failed|FAILED|Error" $O/gpu_tests.log | tail -5
for i in 1 2 3; do MCPT_LIB=$PWD/monte-carlo-path-tracing_amd/exp/hooks/libmcpt_hip.so timeout 300 python tools/experiments/wave_timeline.py dragon --out $O/timeline_dragon_$i.json > $O/timeline_dragon_$i.log 2>&1; done
MCPT_LIB=$PWD/monte-carlo-path-tracing_amd/exp/hooks/libmcpt_hip.so timeout 300 python tools/experiments/wave_timeline.py dragon --share 8 --out $O/timeline_dragon_share8.json > $O/timeline_dragon_share8.log 2>&1
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_s22/timeline_*.json')):
    d=json.load(open(f)); print(f.split('/')[-1], round(d['kernel_ms'],1), round(d['mean_occupancy_of_the_launch'],3), d['alive_at_twentieths_of_the_frame'])
P
timeout 900 python bench.py --workload dragon --steps 10 > $O/bench_dragon.json 2> $O/bench_dragon.err; tail -c 700 $O/bench_dragon.json
