#!/bin/bash
# round 6, GPU session 15: how full is the GPU over dragon's frame on the final kernel (wavefront clocks, measurement-hooks build), three draws
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r06_s15; mkdir -p $O
export MCPT_LIB=$PWD/monte-carlo-path-tracing_amd/exp/hooks/libmcpt_hip.so
for i in 1 2 3; do timeout 300 python tools/experiments/wave_timeline.py dragon --out $O/timeline_dragon_$i.json > $O/timeline_dragon_$i.log 2>&1; tail -c 1500 $O/timeline_dragon_$i.log; done
for s in 8; do timeout 300 python tools/experiments/wave_timeline.py dragon --share $s --out $O/timeline_dragon_share$s.json > $O/timeline_dragon_share$s.log 2>&1; tail -c 1200 $O/timeline_dragon_share$s.log; done
timeout 300 python tools/experiments/wave_timeline.py cornell --out $O/timeline_cornell.json > $O/timeline_cornell.log 2>&1; tail -c 1200 $O/timeline_cornell.log
