#!/bin/bash
# round 6, GPU session 64: dragon's unit without machine LICM (csrc/Makefile, EXPERIMENTS R6-21) — the GPU suite on the new library, dragon's and
# the default bench records, 20 draws, smoke, the kernel-trace summary of the default command
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r06_s64; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q > $O/gpu_suite.log 2>&1; echo "pytest rc=$?" >> $O/gpu_suite.log
grep -E "passed|failed|pytest rc" $O/gpu_suite.log | tail -3
timeout 900 python bench.py --workload dragon --steps 5 > $O/bench_dragon.line.json 2> $O/bench_dragon.err; cut -c1-300 $O/bench_dragon.line.json
cp gpurun_out/bench_detail_dragon_n1.json $O/
timeout 900 python bench.py > $O/bench_default_line.json 2> $O/bench_default.err; wc -c $O/bench_default_line.json
cp gpurun_out/bench_detail_cornell_n1.json $O/
timeout 300 python tools/experiments/market_counts.py dragon 20 > $O/dragon_20_draws.json 2>> $O/err.log; cut -c1-200 $O/dragon_20_draws.json
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-pmc --no-throughput-mode > $GRAFT_REPO_ROOT/$O/prof_stdout.txt 2>&1)
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -r head -4 | cut -c1-160
