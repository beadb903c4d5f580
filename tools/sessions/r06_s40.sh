#!/bin/bash
# round 6, GPU session 40: cornell with the camera-ray pre-pass AND the cost layout (MCPT_LDS_PREPASS_LAYOUT=1, hooks build): does the next sample in the same step pay for the pre-pass?
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r06_s40; mkdir -p $O
export MCPT_LIB=$PWD/monte-carlo-path-tracing_amd/exp/hooks/libmcpt_hip.so
MCPT_LDS_PREPASS_LAYOUT=1 timeout 600 python tools/experiments/prepass_ab.py cornell 8 > $O/prepass_cornell_layout.jsonl 2> $O/err.log; cut -c1-700 $O/prepass_cornell_layout.jsonl; tail -2 $O/err.log
