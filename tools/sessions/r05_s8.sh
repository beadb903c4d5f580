#!/bin/bash
# round 5, GPU session 8: lanes per path by tile cost — kappa sweep (matpreview unit at 4 wavefronts per SIMD: exp/w4)
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp MCPT_COST_DEBUG=1 MCPT_LIB=$PWD/monte-carlo-path-tracing_amd/exp/w4/libmcpt_hip.so
O=gpurun_out/r05_s8; mkdir -p $O
run() { n=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  echo "== $n" >> $O/err.log
  env "${envs[@]}" timeout 300 python tools/experiments/wave_timeline.py "$@" --out $O/$n.json 2>> $O/err.log | cut -c1-100; }
run rc_levels0 MCPT_LEVELS=0 -- matpreview-rc
for k in 0.6 0.4 0.3 0.2 0.12; do run rc_k$k MCPT_LEVEL_KAPPA=$k -- matpreview-rc; done
run rd_levels0 MCPT_LEVELS=0 -- matpreview-rd
for k in 0.4 0.3 0.2; do run rd_k$k MCPT_LEVEL_KAPPA=$k -- matpreview-rd; done
for k in 0.45 0.35 0.25 0.15; do run dragon_s1_k$k MCPT_COST_ORDER=4 MCPT_LEVEL_KAPPA=$k -- dragon --spread 1; done
for k in 0.35 0.2; do run dragon_s2_k$k MCPT_COST_ORDER=4 MCPT_LEVEL_KAPPA=$k -- dragon; done
run dragon_default MCPT_LEVELS=0 -- dragon
grep -E "4096 resident.*of (1048576|921600)|3072 resident.*of 921600|^==" $O/err.log
