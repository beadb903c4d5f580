#!/bin/bash
# round 6, GPU session 50: lanes per path by tile cost re-fitted on the final kernels (kappa sweep, matpreview; hooks build)
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r06_s50; mkdir -p $O
export MCPT_LIB=$PWD/monte-carlo-path-tracing_amd/exp/hooks/libmcpt_hip.so
for k in 0.6 0.45 0.35 0.8 1.0 0.6; do
  for w in matpreview-rc matpreview-rd; do
    MCPT_LEVEL_KAPPA=$k timeout 300 python tools/experiments/market_counts.py $w 6 2>> $O/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); d['kappa']='$k'; print(json.dumps({k:d[k] for k in ('workload','kappa','median_ms','min_ms','max_ms')}))" >> $O/kappa.jsonl
    tail -1 $O/kappa.jsonl
  done
done
