#!/bin/bash
# round 6, GPU session 43: the committed form (escaped lanes first in the diffuse instantiations only): the GPU suite, frame times against session 37's library, dragon's bench line
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r06_s43; mkdir -p $O
timeout 1800 python -m pytest tests/ -m gpu -q > $O/gpu_tests.log 2>&1
grep -E "passed|failed|FAILED|Error" $O/gpu_tests.log | tail -8
L=monte-carlo-path-tracing_amd
timeout 1500 python tools/ab_libraries.py --workloads matpreview-rc,matpreview-rd,dragon --draws 6 --rounds 2 final=$L/libmcpt_hip.so before=$L/exp/before_split/libmcpt_hip.so > $O/ab.jsonl 2> $O/ab.err
cut -c1-300 $O/ab.jsonl; tail -2 $O/ab.err
timeout 900 python bench.py --workload dragon --steps 10 > $O/bench_dragon.json 2> $O/bench_dragon.err; tail -c 600 $O/bench_dragon.json
timeout 900 python bench.py > $O/bench_default_line.json 2> $O/bench_default.err; wc -c $O/bench_default_line.json
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
