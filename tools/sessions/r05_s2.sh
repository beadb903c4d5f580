#!/bin/bash
# round 5, GPU session 2: the quantised 64-byte node records of the pool walk (MCPT_POOL_QUANT, in-tree build) — parity
# tests first, then A/B against the exact 128-byte records on the three mesh workloads.
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r05_s2; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_units.py tests/test_baseline_configs.py -m gpu -x -q -k "pool_walk_answers or intersection_records or reduced_film or small_spp or dragon_full_film or walk_self_check" > $O/parity.log 2>&1
tail -5 $O/parity.log
E=monte-carlo-path-tracing_amd/exp
timeout 1500 python tools/ab_libraries.py --workloads dragon,matpreview-rc,matpreview-rd --draws 3 --rounds 2 \
  exact_f=$E/exact_f/libmcpt_hip.so quant_f_noalign=$E/quant_f/libmcpt_hip.so@MCPT_WIDE_ALIGN=0 quant_f=$E/quant_f/libmcpt_hip.so quant=monte-carlo-path-tracing_amd/libmcpt_hip.so \
  > $O/ab_quant.json 2> $O/ab_quant.err
cat $O/ab_quant.json
