#!/bin/bash
# round 6, GPU session 56 (closing, on the committed tree): the GPU suite, every BASELINE workload's bench record, the kernel-trace summary of the
# driver's default command, the default line itself, smoke
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
bash tools/gpu_session_bench.sh r06_s56
O=gpurun_out/r06_s56
timeout 900 python bench.py > $O/bench_default_line.json 2> $O/bench_default.err; wc -c $O/bench_default_line.json; cut -c1-400 $O/bench_default_line.json
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
cp gpurun_out/bench_detail_*_n1.json $O/ 2>/dev/null
