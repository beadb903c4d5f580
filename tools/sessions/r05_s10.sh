#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r05_s10; mkdir -p $O
timeout 1500 python -m pytest tests/ -m gpu -q > $O/gpu_tests.log 2>&1
tail -8 $O/gpu_tests.log
