#!/bin/bash
# round 6, GPU session 39 (final library, after the in-step regeneration): the GPU suite, the bench lines with counters (five workloads), the kernel-trace summary of the
# default bench command, rank shares on one GPU, full-size parity of the five configurations
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
STEPS=10 bash tools/gpu_session_bench.sh r06_s39
O=gpurun_out/r06_s39
timeout 900 python bench.py > $O/bench_default_line.json 2> $O/bench_default.err; cat $O/bench_default_line.json | cut -c1-600
timeout 1200 python tools/experiments/strong_share.py > $O/strong_share.log 2>&1; cp gpurun_out/strong_share.json $O/strong_share.json; tail -3 $O/strong_share.log | cut -c1-300
timeout 1500 python tests/full_size_parity.py > $O/full_size_parity.log 2>&1; cp gpurun_out/full_size_parity.json $O/full_size_parity.json; tail -5 $O/full_size_parity.log | cut -c1-300
