#!/bin/bash
# round 5, GPU session 9: the whole GPU suite + the default bench with the round's scheduling changes
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r05_s9; mkdir -p $O
timeout 1500 python -m pytest tests/ -m gpu -x -q > $O/gpu_tests.log 2>&1
tail -5 $O/gpu_tests.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
cat $O/bench_default.json
for w in matpreview-rc matpreview-rd volumetric; do
  timeout 600 python bench.py --workload $w --no-pmc --no-throughput-mode --steps 3 > $O/bench_$w.json 2> $O/bench_$w.err; cut -c1-300 $O/bench_$w.json
done
