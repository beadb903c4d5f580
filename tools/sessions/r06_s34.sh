#!/bin/bash
# round 6, GPU session 34: back-end flags on the class-sorted unit (volumetric-caustic)
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r06_s34; mkdir -p $O
L=monte-carlo-path-tracing_amd
timeout 1500 python tools/ab_libraries.py --workloads volumetric --draws 5 --rounds 2 prod=$L/libmcpt_hip.so max_ilp=$L/exp/v_ilp/libmcpt_hip.so max_memory_clause=$L/exp/v_memclause/libmcpt_hip.so prealloc=$L/exp/v_prealloc/libmcpt_hip.so > $O/ab.jsonl 2> $O/ab.err
cut -c1-200 $O/ab.jsonl; tail -2 $O/ab.err
