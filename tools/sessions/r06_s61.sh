#!/bin/bash
# round 6, GPU session 61: the full-size parity tests (MCPT_FULL_PARITY=1: the five BASELINE configurations at full film AND full spp against the
# oracle on the box's host threads) on the committed tree
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r06_s61; mkdir -p $O
MCPT_FULL_PARITY=1 timeout 2400 python -m pytest tests -m "gpu and full_parity" -q > $O/full_parity_suite.log 2>&1
echo "pytest rc=$?" >> $O/full_parity_suite.log
grep -E "passed|failed|pytest rc" $O/full_parity_suite.log | tail -3
