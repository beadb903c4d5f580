#!/bin/bash
# round 6, GPU session 12: records read through a pointer to the kernel-argument segment (MCPT_SCENE_POINTER=2) against a device buffer (=1) and by value;
# volumetric-caustic's exchange in one pass of 32 words (two barriers a step, 16 KB) against two passes of 16
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r06_s12; mkdir -p $O
L=monte-carlo-path-tracing_amd
timeout 2000 python tools/ab_libraries.py --workloads cornell,volumetric,matpreview-rc,matpreview-rd,dragon --draws 6 --rounds 2 by_value=$L/libmcpt_hip.so pointer=$L/exp/sceneptr/libmcpt_hip.so kernarg=$L/exp/kernargptr/libmcpt_hip.so > $O/ab.jsonl 2> $O/ab.err
cut -c1-200 $O/ab.jsonl; tail -3 $O/ab.err
timeout 600 python tools/ab_libraries.py --workloads volumetric --draws 6 --rounds 2 two_passes=$L/libmcpt_hip.so one_pass=$L/exp/sort1pass/libmcpt_hip.so > $O/ab_sort.jsonl 2> $O/ab_sort.err
cut -c1-200 $O/ab_sort.jsonl
