#!/bin/bash
# round 6, GPU session 32: the committed library — two renderers at once (the market's residency guard), the GPU suite, smoke, the default bench line, dragon's throughput mode
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r06_s32; mkdir -p $O
timeout 420 python -m pytest tests/test_baseline_configs.py -m gpu -q -s -k "two_renderers" > $O/two_renderers.log 2>&1; tail -5 $O/two_renderers.log | cut -c1-400
timeout 1800 python -m pytest tests/ -m gpu -q > $O/gpu_tests.log 2>&1
grep -E "passed|failed|FAILED|Error" $O/gpu_tests.log | tail -8
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 900 python bench.py > $O/bench_default_line.json 2> $O/bench_default.err; wc -c $O/bench_default_line.json; cut -c1-700 $O/bench_default_line.json
timeout 600 python tools/ab_libraries.py --workloads dragon --rng 1 --draws 6 --rounds 1 final=monte-carlo-path-tracing_amd/libmcpt_hip.so > $O/dragon_pcg.jsonl 2> $O/dragon_pcg.err; cut -c1-300 $O/dragon_pcg.jsonl
