#!/bin/bash
# round 6, GPU session 42: resolve with the escaped lanes first and ONE common surface resolve (path_core.h resolve_and_regenerate) against the form of session 37: parity tests, frame times
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r06_s42; mkdir -p $O
timeout 1800 python -m pytest tests/ -m gpu -q > $O/gpu_tests.log 2>&1
grep -E "passed|failed|FAILED|Error" $O/gpu_tests.log | tail -8
L=monte-carlo-path-tracing_amd
timeout 1500 python tools/ab_libraries.py --workloads matpreview-rc,matpreview-rd,dragon --draws 6 --rounds 2 escaped_first=$L/libmcpt_hip.so before=$L/exp/before_split/libmcpt_hip.so > $O/ab.jsonl 2> $O/ab.err
cut -c1-300 $O/ab.jsonl; tail -2 $O/ab.err
