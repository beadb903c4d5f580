#!/bin/bash
# round 6, GPU session 23: the GPU suite with dragon's unit compiled without the SLP vectoriser (the tail spread's pending shadow ray: R6-1's defect again); dragon's frame time
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r06_s23; mkdir -p $O
timeout 1800 python -m pytest tests/ -m gpu -q > $O/gpu_tests.log 2>&1
grep -E "passed|failed|FAILED|Error" $O/gpu_tests.log | tail -8
timeout 600 python tools/ab_libraries.py --workloads dragon --draws 12 --rounds 2 final=monte-carlo-path-tracing_amd/libmcpt_hip.so > $O/dragon.jsonl 2> $O/dragon.err
cut -c1-400 $O/dragon.jsonl
