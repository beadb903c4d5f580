#!/bin/bash
# round 6, GPU session 13: the GPU suite on the library with the records behind a pointer (production form: a ring of device slots) and its A/B against -DMCPT_SCENE_POINTER=0
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r06_s13; mkdir -p $O
L=monte-carlo-path-tracing_amd
timeout 1500 python -m pytest tests/ -m gpu -x -q > $O/gpu_tests.log 2>&1
tail -5 $O/gpu_tests.log
timeout 1500 python tools/ab_libraries.py --workloads cornell,volumetric,matpreview-rc,matpreview-rd --draws 6 --rounds 2 pointer=$L/libmcpt_hip.so by_value=$L/exp/byvalue/libmcpt_hip.so > $O/ab.jsonl 2> $O/ab.err
cut -c1-200 $O/ab.jsonl; tail -3 $O/ab.err
