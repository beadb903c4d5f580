#!/bin/bash
# round 5, measurement session: GPU suite, the driver's default line + its kernel trace, the five workload lines with counters,
# rank shares, dragon's frame time over 10 processes x 2 draws, full-film full-spp parity of all five configurations
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r05_final; mkdir -p $O
timeout 1500 python -m pytest tests/ -m gpu -q > $O/gpu_tests.log 2>&1
grep -E "passed|failed|FAILED" $O/gpu_tests.log | tail -3
( time timeout 900 python bench.py > $O/bench_default.line 2> $O/bench_default.err ) 2> $O/bench_default.time
cat $O/bench_default.line; wc -c $O/bench_default.line; tail -3 $O/bench_default.time
cp gpurun_out/bench_detail_cornell_n1.json $O/bench_default_detail.json
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o ks -- python $OLDPWD/bench.py --no-cpu-baseline --no-pmc --no-throughput-mode > /dev/null 2>&1 )
cp $(find /tmp/prof -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
for w in cornell dragon matpreview-rc matpreview-rd volumetric; do
  timeout 900 python bench.py --workload $w --steps 3 --warmup 1 --no-also > $O/bench_$w.line 2> $O/bench_$w.err
  echo "bench $w rc=$?"; cp gpurun_out/bench_detail_${w}_n1.json $O/bench_$w.json; head -c 400 $O/bench_$w.line; echo
done
timeout 900 python tools/experiments/strong_share.py > $O/strong_share.log 2>&1; cp gpurun_out/strong_share.json $O/strong_share.json
timeout 900 python tools/ab_libraries.py --workloads dragon,cornell --draws 2 --rounds 10 lib=monte-carlo-path-tracing_amd/libmcpt_hip.so > $O/spread_n20.json 2> $O/spread_n20.err
cat $O/spread_n20.json
timeout 1500 python tests/full_size_parity.py > $O/full_size_parity.log 2>&1; echo "full parity rc=$?"; cp gpurun_out/full_size_parity.json $O/full_size_parity.json
grep -o '"config": "[^"]*"\|"frac_bit_exact": [0-9.]*\|"hip_msamples_per_s": [0-9.]*' $O/full_size_parity.log | paste - - - | cut -c1-200
