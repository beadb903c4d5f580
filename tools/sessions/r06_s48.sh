#!/bin/bash
# round 6, GPU session 48: round 5's library (built from commit c6f18fa) and this round's, side by side on one box, the five BASELINE workloads
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r06_s48; mkdir -p $O
L=monte-carlo-path-tracing_amd
timeout 2400 python tools/ab_libraries.py --workloads cornell,dragon,matpreview-rc,matpreview-rd,volumetric --draws 6 --rounds 2 round5=$L/exp/round5/libmcpt_hip.so round6=$L/libmcpt_hip.so > $O/ab.jsonl 2> $O/ab.err
cut -c1-200 $O/ab.jsonl; tail -2 $O/ab.err
