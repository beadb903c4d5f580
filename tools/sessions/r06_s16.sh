#!/bin/bash
# round 6, GPU session 16: dragon — hand-out most expensive first with lanes per path by tile cost, re-measured on the final kernel
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r06_s16; mkdir -p $O
export MCPT_LIB=$PWD/monte-carlo-path-tracing_amd/exp/hooks/libmcpt_hip.so
MCPT_COST_DEBUG=1 timeout 1500 python tools/experiments/dragon_cost_order_levels.py dragon 12 > $O/dragon_cost_order.jsonl 2> $O/dragon_cost_order.err
cut -c1-330 $O/dragon_cost_order.jsonl; grep "lanes per path" $O/dragon_cost_order.err | head
