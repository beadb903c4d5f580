#!/bin/bash
# round 6, closing check of the committed tree: smoke, the GPU suite, the driver's default bench command
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r06_final; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 1800 python -m pytest tests/ -m gpu -q > $O/gpu_tests.log 2>&1
grep -E "passed|failed|FAILED|Error" $O/gpu_tests.log | tail -5
timeout 900 python bench.py > $O/bench_default_line.json 2> $O/bench_default.err; wc -c $O/bench_default_line.json; cut -c1-400 $O/bench_default_line.json
