#!/bin/bash
# round 6, GPU session 59: XCD bands (an XCD's L2 sees one stripe of the image) WITH the tail spread and the path market — R6-5 measured the bands
# before the market existed and lost 9 % to their tails
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r06_s59; mkdir -p $O
X=$PWD/monte-carlo-path-tracing_amd/exp/bands_market/libmcpt_hip.so
for round in 1 2; do
  timeout 300 python tools/experiments/market_counts.py dragon 10 2>> $O/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); d['arm']='production'; print(json.dumps(d))" >> $O/bands.jsonl
  MCPT_TILE_ORDER=2 timeout 300 python tools/experiments/market_counts.py dragon 10 2>> $O/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); d['arm']='bands, no market (production library)'; print(json.dumps(d))" >> $O/bands.jsonl
  MCPT_LIB=$X MCPT_TILE_ORDER=2 timeout 300 python tools/experiments/market_counts.py dragon 10 2>> $O/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); d['arm']='bands + tail spread + market'; print(json.dumps(d))" >> $O/bands.jsonl
done
python -c "
import json
for l in open('gpurun_out/r06_s59/bands.jsonl'):
    d=json.loads(l); print(d['arm'], d['median_ms'], d['min_ms'], d['max_ms'], d['market_tickets_given_finished'], d['sha'], d['kernel'][:100])
"
tail -3 $O/err.log
