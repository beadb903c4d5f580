#!/bin/bash
# round 5, GPU session 7: lanes per path by tile cost (RenderJob::level_until) — sweeps
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp MCPT_COST_DEBUG=1
O=gpurun_out/r05_s7; mkdir -p $O
run() { # name, env..., -- args
  n=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  echo "== $n" >> $O/err.log
  env "${envs[@]}" timeout 300 python tools/experiments/wave_timeline.py "$@" --out $O/$n.json 2>> $O/err.log | cut -c1-400
}
run rc_levels0 MCPT_LEVELS=0 -- matpreview-rc
run rc_k060 MCPT_LEVEL_KAPPA=0.6 -- matpreview-rc
run rc_k045 MCPT_LEVEL_KAPPA=0.45 -- matpreview-rc
run rc_k080 MCPT_LEVEL_KAPPA=0.8 -- matpreview-rc
run rd_levels0 MCPT_LEVELS=0 -- matpreview-rd
run rd_k060 MCPT_LEVEL_KAPPA=0.6 -- matpreview-rd
run rd_k045 MCPT_LEVEL_KAPPA=0.45 -- matpreview-rd
run dragon_default MCPT_LEVELS=0 -- dragon
run dragon_ordered_levels0 MCPT_LEVELS=0 MCPT_COST_ORDER=4 -- dragon
run dragon_ordered_k060 MCPT_COST_ORDER=4 -- dragon
run dragon_ordered_k060_spread1 MCPT_COST_ORDER=4 -- dragon --spread 1
run dragon_ordered_k045_spread1 MCPT_COST_ORDER=4 MCPT_LEVEL_KAPPA=0.45 -- dragon --spread 1
run dragon_ordered_k080_spread1 MCPT_COST_ORDER=4 MCPT_LEVEL_KAPPA=0.8 -- dragon --spread 1
grep -E "lanes per path|tile costs|^==" $O/err.log
timeout 900 python -m pytest tests/test_baseline_configs.py -m gpu -x -q -k "reduced_film or small_spp or dragon_full_film" > $O/parity.log 2>&1
tail -3 $O/parity.log
