#!/bin/bash
# round 6, GPU session 62: draw-to-draw spread of the final library, 20 draws of every BASELINE workload (VERDICT round 5 task 2: dragon's median of
# 20 draws with a min-max spread <= 6 %)
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r06_s62; mkdir -p $O
for w in cornell dragon matpreview-rc matpreview-rd volumetric; do
  timeout 600 python tools/experiments/market_counts.py $w 20 >> $O/spread20.jsonl 2>> $O/err.log
  tail -1 $O/spread20.jsonl | cut -c1-170
done
