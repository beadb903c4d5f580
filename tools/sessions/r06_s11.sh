#!/bin/bash
# round 6, GPU session 11: the scene and the job behind one pointer (-DMCPT_SCENE_POINTER=1) against the by-value kernel arguments
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r06_s11; mkdir -p $O
L=monte-carlo-path-tracing_amd
timeout 1700 python tools/ab_libraries.py --workloads cornell,matpreview-rc,matpreview-rd,volumetric,dragon --draws 6 --rounds 2 by_value=$L/libmcpt_hip.so pointer=$L/exp/sceneptr/libmcpt_hip.so > $O/ab.jsonl 2> $O/ab.err
cut -c1-300 $O/ab.jsonl; tail -3 $O/ab.err
