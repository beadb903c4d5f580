#!/bin/bash
# round 5, GPU session 4: how full is the GPU over a frame (wave start / end clocks)
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r05_s4; mkdir -p $O
for spec in "dragon" "dragon --spread 1" "dragon --spread 4" "dragon --share 8" "matpreview-rd" "matpreview-rc" "cornell" "volumetric"; do
  n=$(echo $spec | tr ' ' '_' | tr -d '-')
  timeout 300 python tools/experiments/wave_timeline.py $spec --out $O/timeline_$n.json 2>> $O/err.log
done
