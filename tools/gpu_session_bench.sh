#!/bin/bash
# One GPU-box session: the -m gpu suite, then bench.py for every BASELINE workload (each line carries its own
# roofline incl. the rocprofv3 counter passes and the CPU baseline), then the kernel-trace summary of the
# default bench command.  Outputs under gpurun_out/$TAG/.
TAG=${1:-r02}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
  echo "pytest rc=$?" >> $OUT/pytest_gpu.log
  tail -3 $OUT/pytest_gpu.log
fi
for w in ${WORKLOADS:-cornell dragon matpreview-rc matpreview-rd volumetric}; do
  timeout 900 python bench.py --workload $w --steps ${STEPS:-3} > $OUT/bench_$w.json 2> $OUT/bench_$w.err
  echo "bench $w rc=$?"; tail -c 600 $OUT/bench_$w.json
done
if [ "${SKIP_PROF:-0}" != "1" ]; then
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- \
      python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-pmc --no-throughput-mode > $GRAFT_REPO_ROOT/$OUT/prof_stdout.txt 2>&1)
  find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -r head -5
fi
