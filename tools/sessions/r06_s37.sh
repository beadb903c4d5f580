#!/bin/bash
# round 6, GPU session 37: the next sample in the same step (scenes with the camera-ray pre-pass): parity tests first, then frame times against -DMCPT_REGENERATE_ROUNDS=0
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r06_s37; mkdir -p $O
timeout 1800 python -m pytest tests/ -m gpu -q > $O/gpu_tests.log 2>&1
grep -E "passed|failed|FAILED|Error" $O/gpu_tests.log | tail -8
L=monte-carlo-path-tracing_amd
timeout 1500 python tools/ab_libraries.py --workloads matpreview-rc,matpreview-rd,dragon --draws 6 --rounds 2 same_step=$L/libmcpt_hip.so next_step=$L/exp/noregen/libmcpt_hip.so > $O/ab.jsonl 2> $O/ab.err
cut -c1-300 $O/ab.jsonl; tail -2 $O/ab.err
