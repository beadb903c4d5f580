#!/bin/bash
# round 6, GPU session 21: the tail spread (paths dealt out over a workgroup's four wavefronts once the work counter is dry) against the same library without it
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r06_s21; mkdir -p $O
L=monte-carlo-path-tracing_amd
timeout 900 python tools/ab_libraries.py --workloads dragon --draws 12 --rounds 2 tail_spread=$L/libmcpt_hip.so without=$L/exp/nospread/libmcpt_hip.so > $O/ab_dragon.jsonl 2> $O/ab_dragon.err
cut -c1-500 $O/ab_dragon.jsonl; tail -3 $O/ab_dragon.err
timeout 1200 python tools/ab_libraries.py --workloads matpreview-rc,matpreview-rd --draws 5 --rounds 2 tail_spread=$L/libmcpt_hip.so without=$L/exp/nospread/libmcpt_hip.so > $O/ab_matpreview.jsonl 2> $O/ab_matpreview.err
cut -c1-300 $O/ab_matpreview.jsonl
timeout 900 python tools/ab_libraries.py --workloads dragon,matpreview-rc --share 8 --draws 5 --rounds 2 tail_spread=$L/libmcpt_hip.so without=$L/exp/nospread/libmcpt_hip.so > $O/ab_share8.jsonl 2> $O/ab_share8.err
cut -c1-300 $O/ab_share8.jsonl
