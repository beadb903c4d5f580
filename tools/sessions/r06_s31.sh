#!/bin/bash
# round 6, GPU session 31: the records behind a pointer in the general surface / full-feature pool-walk units, on the reference's other scenes (640 x 360 spp 64)
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r06_s31; mkdir -p $O
L=monte-carlo-path-tracing_amd
timeout 1500 python tools/ab_libraries.py --workloads other:classroom,other:dining-room,other:box,other:matpreview-rough-plastic,other:matpreview-thin-dielectric --draws 6 --rounds 2 by_value=$L/libmcpt_hip.so pointer=$L/exp/ptr_general/libmcpt_hip.so > $O/ab.jsonl 2> $O/ab.err
cut -c1-260 $O/ab.jsonl; tail -2 $O/ab.err
