#!/bin/bash
# round 5, GPU session 14: merged queries outside LDS — the whole GPU suite, timings, shares
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r05_s14; mkdir -p $O
timeout 1500 python -m pytest tests/ -m gpu -q > $O/gpu_tests.log 2>&1
grep -E "passed|failed|FAILED" $O/gpu_tests.log | tail -5
for spec in "cornell" "dragon" "matpreview-rc" "matpreview-rd" "volumetric" "cornell --share 8" "dragon --share 8" "matpreview-rc --share 8"; do
  n=$(echo $spec | tr ' ' '_' | tr -d '-')
  timeout 300 python tools/experiments/wave_timeline.py $spec --out $O/timeline_$n.json 2>> $O/err.log | cut -c1-300
done
