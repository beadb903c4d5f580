#!/bin/bash
# round 5, GPU session 13: merged queries (a vertex's last shadow ray travels with the next segment's closest query) — parity, timing
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r05_s13; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_baseline_configs.py -m gpu -x -q -k "golden_frames or scheduling_choices or pool_walk or reduced_film or small_spp or dragon_full_film or full_size_properties or lane_spread" > $O/parity.log 2>&1
tail -15 $O/parity.log
for spec in "cornell" "dragon" "matpreview-rc" "matpreview-rd" "cornell --share 8" "dragon --share 8"; do
  n=$(echo $spec | tr ' ' '_' | tr -d '-')
  timeout 300 python tools/experiments/wave_timeline.py $spec --out $O/timeline_$n.json 2>> $O/err.log | cut -c1-330
done
