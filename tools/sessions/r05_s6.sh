#!/bin/bash
# round 5, GPU session 6: the work counter hands items out when a lane is free (no reservation ahead): parity, then timelines
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r05_s6; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_baseline_configs.py -m gpu -x -q -k "golden_frames or scheduling_choices or class_sort or reduced_film or small_spp or full_size_properties or tile_hand_out" > $O/parity.log 2>&1
tail -5 $O/parity.log
for spec in "dragon" "dragon --spread 4" "matpreview-rc" "matpreview-rd" "cornell" "volumetric"; do
  n=$(echo $spec | tr ' ' '_' | tr -d '-')
  timeout 300 python tools/experiments/wave_timeline.py $spec --out $O/timeline_$n.json 2>> $O/err.log
done
