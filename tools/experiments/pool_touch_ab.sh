#!/bin/bash
# A/B of the pool walk's child-record touch (MCPT_POOL_TOUCH; EXPERIMENTS R4-8 — the code is no longer in the tree) — separate processes, alternating
mkdir -p gpurun_out/r04d
for w in dragon matpreview-rc matpreview-rd; do
  for t in 0 1 0 1; do
    echo "== $w touch=$t" >> gpurun_out/r04d/touch_ab.log
    MCPT_POOL_TOUCH=$t timeout 300 python tools/ab_draws.py $w --reps 5 default >> gpurun_out/r04d/touch_ab.log 2>&1
  done
done
cat gpurun_out/r04d/touch_ab.log | cut -c1-400
