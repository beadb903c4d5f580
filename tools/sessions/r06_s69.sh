#!/bin/bash
# round 6, GPU session 69: the library with the surface-material / general pool-walk units compiled without the SLP vectoriser (csrc/Makefile): the GPU
# suite, the reference's other scenes against the oracle, the full-size parity tests, matpreview's bench records, the default line, smoke
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r06_s69; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q > $O/gpu_suite.log 2>&1; echo "pytest rc=$?" >> $O/gpu_suite.log
grep -E "passed|failed|pytest rc" $O/gpu_suite.log | tail -3
timeout 900 python tools/experiments/other_scenes_parity.py 640 360 64 > $O/other_scenes_parity.jsonl 2>> $O/err.log; cut -c1-140 $O/other_scenes_parity.jsonl
for w in matpreview-rc matpreview-rd; do
  timeout 900 python bench.py --workload $w --steps 3 > $O/bench_$w.line.json 2> $O/bench_$w.err; cut -c1-200 $O/bench_$w.line.json
  cp gpurun_out/bench_detail_${w}_n1.json $O/
done
timeout 900 python bench.py > $O/bench_default_line.json 2> $O/bench_default.err; wc -c $O/bench_default_line.json; cut -c1-160 $O/bench_default_line.json
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
MCPT_FULL_PARITY=1 timeout 2400 python -m pytest tests -m "gpu and full_parity" -q > $O/full_parity_suite.log 2>&1; echo "pytest rc=$?" >> $O/full_parity_suite.log
grep -E "passed|failed|pytest rc" $O/full_parity_suite.log | tail -3
