#!/bin/bash
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r05_s11; mkdir -p $O
run() { n=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  echo "== $n" >> $O/err.log
  env "${envs[@]}" timeout 300 python tools/experiments/wave_timeline.py "$@" --out $O/$n.json 2>> $O/err.log | cut -c1-100; }
run vol_default X=1 -- volumetric
run vol_pool2 X=1 -- volumetric --pool 2
run vol_nosort X=1 -- volumetric --sort 0
run vol_nosort_pool2 X=1 -- volumetric --sort 0 --pool 2
run rd_k06 X=1 -- matpreview-rd
run rd_levels0 MCPT_LEVELS=0 -- matpreview-rd
run rc_k05 MCPT_LEVEL_KAPPA=0.5 -- matpreview-rc
run rc_k07 MCPT_LEVEL_KAPPA=0.7 -- matpreview-rc
