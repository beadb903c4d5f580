#!/bin/bash
# round 5, GPU session 12: GPU suite + the driver's default line (with the memory-side counter passes) + its kernel trace
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r05_s12; mkdir -p $O
timeout 1500 python -m pytest tests/ -m gpu -q > $O/gpu_tests.log 2>&1
tail -4 $O/gpu_tests.log
( time timeout 900 python bench.py > $O/bench_default.line 2> $O/bench_default.err ) 2> $O/bench_default.time
cat $O/bench_default.line; wc -c $O/bench_default.line; cat $O/bench_default.time
cp gpurun_out/bench_detail_cornell_n1.json $O/bench_default_detail.json
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o ks -- python $OLDPWD/bench.py --no-cpu-baseline --no-pmc --no-throughput-mode > /dev/null 2>&1; cd $OLDPWD
cp $(find /tmp/prof -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv; head -8 $O/bench_kernel_stats.csv | cut -c1-200
