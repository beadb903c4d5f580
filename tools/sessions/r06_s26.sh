#!/bin/bash
# round 6, GPU session 26: the path market with the tiles handed out most expensive first (measurement-hooks build, MCPT_COST_ORDER=4)
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r06_s26; mkdir -p $O
export MCPT_LIB=$PWD/monte-carlo-path-tracing_amd/exp/hooks/libmcpt_hip.so
for s in "" "MCPT_COST_ORDER=4 MCPT_LEVELS=0" "MCPT_COST_ORDER=4 MCPT_LEVEL_KAPPA=0.8" "MCPT_COST_ORDER=4 MCPT_LEVEL_KAPPA=0.6" ""; do
  env $s timeout 600 python tools/experiments/market_counts.py dragon 16 2>> $O/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); d['setting']='$s'; print(json.dumps(d))" >> $O/market_cost_order.jsonl
  tail -1 $O/market_cost_order.jsonl | cut -c1-330
done
