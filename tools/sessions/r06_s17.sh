#!/bin/bash
# round 6, GPU session 17: dragon's slowest tile alone, with 1 ... 64 lanes per path
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r06_s17; mkdir -p $O
timeout 1500 python tools/experiments/dragon_slowest_tile.py dragon > $O/dragon_slowest_tile.jsonl 2> $O/dragon_slowest_tile.err
cat $O/dragon_slowest_tile.jsonl; tail -3 $O/dragon_slowest_tile.err
