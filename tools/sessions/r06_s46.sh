#!/bin/bash
# round 6, GPU session 46: the next sample in the same step in the general surface / full-feature pool-walk units, on the reference's other scenes
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r06_s46; mkdir -p $O
L=monte-carlo-path-tracing_amd
timeout 1500 python tools/ab_libraries.py --workloads other:classroom,other:dining-room,other:box,other:matpreview-rough-plastic,other:matpreview-thin-dielectric --draws 6 --rounds 2 same_step=$L/libmcpt_hip.so next_step=$L/exp/noregen_general/libmcpt_hip.so > $O/ab.jsonl 2> $O/ab.err
cut -c1-200 $O/ab.jsonl; tail -2 $O/ab.err
