#!/bin/bash
# round 6, GPU session 14: back-end flags on cornell's unit (render_variants_lean_pool) with the records behind a pointer
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r06_s14; mkdir -p $O
L=monte-carlo-path-tracing_amd
libs="prod=$L/libmcpt_hip.so"
for n in f_ilp f_memclause f_sink f_prealloc; do libs="$libs $n=$L/exp/$n/libmcpt_hip.so"; done
timeout 1500 python tools/ab_libraries.py --workloads cornell --draws 8 --rounds 2 $libs > $O/ab.jsonl 2> $O/ab.err
cut -c1-200 $O/ab.jsonl; tail -3 $O/ab.err
