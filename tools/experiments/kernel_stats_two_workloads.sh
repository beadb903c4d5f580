cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/ks
for w in matpreview-rd volumetric; do
 (cd /tmp && timeout 170 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/ks/prof_$w -o k -- python /root/repo/bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-throughput-mode --no-also > /root/repo/gpurun_out/ks/$w.out 2>&1)
 find gpurun_out/ks/prof_$w -name "*kernel_stats.csv" | head -1 | xargs -r -I{} cp {} gpurun_out/ks/kernel_stats_$w.csv; rm -rf gpurun_out/ks/prof_$w
 head -3 gpurun_out/ks/kernel_stats_$w.csv | cut -c1-200
done
