#!/bin/bash
# round 6, GPU session 57: lanes per path of dragon's launch re-measured on the FINAL kernel (R6-14 measured 1 / 2 / 4 before the next sample in
# the same step, R6-17)
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r06_s57; mkdir -p $O
for s in rule 1 2 4 rule 1; do
  if [ $s = rule ]; then timeout 300 python tools/experiments/market_counts.py dragon 10 >> $O/spread.jsonl 2>> $O/err.log
  else MCPT_SPREAD=$s timeout 300 python tools/experiments/market_counts.py dragon 10 >> $O/spread.jsonl 2>> $O/err.log; fi
  tail -1 $O/spread.jsonl | cut -c1-200
done
