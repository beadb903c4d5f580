#!/bin/bash
# round 6, GPU session 60: what the cost order is worth on the final dragon kernel (image order, with the tail spread and the market) — the
# yardstick for session 59's bands (image order inside every band)
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r06_s60; mkdir -p $O
X=$PWD/monte-carlo-path-tracing_amd/exp/bands_market/libmcpt_hip.so
for round in 1 2; do
  MCPT_TILE_ORDER=0 timeout 300 python tools/experiments/market_counts.py dragon 10 2>> $O/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); d['arm']='image order + tail spread + market (production library)'; print(json.dumps(d))" >> $O/order.jsonl
  MCPT_LIB=$X MCPT_TILE_ORDER=2 timeout 300 python tools/experiments/market_counts.py dragon 10 2>> $O/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); d['arm']='bands + tail spread + market'; print(json.dumps(d))" >> $O/order.jsonl
  timeout 300 python tools/experiments/market_counts.py dragon 10 2>> $O/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); d['arm']='production'; print(json.dumps(d))" >> $O/order.jsonl
done
python -c "
import json
for l in open('gpurun_out/r06_s60/order.jsonl'):
    d=json.loads(l); print(d['arm'], d['median_ms'], d['min_ms'], d['max_ms'], d['market_tickets_given_finished'], d['sha'], d['kernel'][80:160])
"
