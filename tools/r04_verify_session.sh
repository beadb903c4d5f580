cd /root/repo; export TMPDIR=/tmp; OUT=gpurun_out/r04b; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/gpu_tests.log
( time timeout 600 python bench.py > $OUT/bench_default.line 2> $OUT/bench_default.err ) 2> $OUT/bench_default.time; echo "bench rc=$?"
cp gpurun_out/bench_detail_cornell_n1.json $OUT/bench_default.json; wc -c $OUT/bench_default.line; cat $OUT/bench_default.line; tail -3 $OUT/bench_default.time
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$OUT/prof -o bench -- python /root/repo/bench.py --no-cpu-baseline --no-pmc --no-throughput-mode > /root/repo/$OUT/prof_stdout.txt 2>&1)
find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} $OUT/bench_kernel_stats.csv
head -6 $OUT/bench_kernel_stats.csv | cut -c1-220; rm -rf $OUT/prof
timeout 120 python __graft_entry__.py smoke 2>&1 | tail -2
