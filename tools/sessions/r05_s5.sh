#!/bin/bash
# round 5, GPU session 5: frame timelines with the last-pixel-taken clock
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r05_s5; mkdir -p $O
for spec in "dragon" "dragon --spread 4" "dragon --spread 8" "matpreview-rc" "matpreview-rc --spread 2" "matpreview-rd" "matpreview-rd --spread 2" "cornell"; do
  n=$(echo $spec | tr ' ' '_' | tr -d '-')
  timeout 300 python tools/experiments/wave_timeline.py $spec --out $O/timeline_$n.json 2>> $O/err.log
done
