#!/bin/bash
# round 6, GPU session 51: the tail spread and the path market compiled into matpreview's one-BSDF units (-DMCPT_TAIL_SPREAD=2), with and without lanes per path by tile cost
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r06_s51; mkdir -p $O
X=$PWD/monte-carlo-path-tracing_amd/exp/spread_all/libmcpt_hip.so
for w in matpreview-rc matpreview-rd; do
  timeout 300 python tools/experiments/market_counts.py $w 6 2>> $O/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); d['setting']='production'; print(json.dumps(d))" >> $O/spread_all.jsonl; tail -1 $O/spread_all.jsonl | cut -c1-260
  MCPT_LIB=$X timeout 300 python tools/experiments/market_counts.py $w 6 2>> $O/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); d['setting']='market + lanes per path'; print(json.dumps(d))" >> $O/spread_all.jsonl; tail -1 $O/spread_all.jsonl | cut -c1-260
  MCPT_LIB=$X MCPT_LEVELS=0 timeout 300 python tools/experiments/market_counts.py $w 6 2>> $O/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); d['setting']='market, no lanes per path'; print(json.dumps(d))" >> $O/spread_all.jsonl; tail -1 $O/spread_all.jsonl | cut -c1-260
done
