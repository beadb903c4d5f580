#!/bin/bash
# round-4 measurement session on one GPU box: bench lines of the five workloads (each with its counter passes and CPU
# baseline), the driver's default command, the kernel-trace summary of it, rank-share timings, full-film parity.
cd /root/repo
OUT=${OUT:-gpurun_out/r04}
mkdir -p $OUT
export TMPDIR=/tmp
for w in cornell dragon matpreview-rc matpreview-rd volumetric; do
  timeout 600 python bench.py --workload $w --steps 3 --warmup 1 --no-also > $OUT/bench_$w.line 2> $OUT/bench_$w.err
  cp gpurun_out/bench_detail_${w}_n1.json $OUT/bench_$w.json
  echo "bench $w rc=$?"; head -c 300 $OUT/bench_$w.line; echo
done
( time timeout 900 python bench.py > $OUT/bench_default.line 2> $OUT/bench_default.err ) 2> $OUT/bench_default.time
cp gpurun_out/bench_detail_cornell_n1.json $OUT/bench_default.json
wc -c $OUT/bench_default.line; tail -3 $OUT/bench_default.time
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$OUT/prof -o bench -- \
    python /root/repo/bench.py --no-cpu-baseline --no-pmc --no-throughput-mode > /root/repo/$OUT/prof_stdout.txt 2>&1)
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} $OUT/bench_kernel_stats.csv
head -6 $OUT/bench_kernel_stats.csv | cut -c1-200
rm -rf $OUT/prof
timeout 900 python tools/experiments/strong_share.py > $OUT/strong_share.log 2>&1; cp gpurun_out/strong_share.json $OUT/strong_share.json; tail -4 $OUT/strong_share.log | cut -c1-250
timeout 1500 python tests/full_size_parity.py > $OUT/full_size_parity.log 2>&1; cp gpurun_out/full_size_parity.json $OUT/full_size_parity.json; grep -o '"config": "[^"]*"\|"frac_bit_exact": [0-9.]*\|"hip_msamples_per_s": [0-9.]*' $OUT/full_size_parity.log | paste - - - | cut -c1-200
