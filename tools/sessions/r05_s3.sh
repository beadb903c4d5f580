#!/bin/bash
# round 5, GPU session 3: child-parallel node steps of the pool walk (MCPT_POOL_CHILD_PARALLEL; exp/cp) — parity of the pool-walk
# kernels first, then A/B against one lane per item: whole frames and rank 0's 1/8 share.
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp MCPT_WIDE_ALIGN=0
O=gpurun_out/r05_s3; mkdir -p $O
E=monte-carlo-path-tracing_amd/exp
MCPT_LIB=$PWD/$E/cp/libmcpt_hip.so timeout 1200 python -m pytest tests/test_gpu_units.py tests/test_gpu_parity.py tests/test_baseline_configs.py -m gpu -x -q -k "pool_walk or intersection_records or reduced_film or golden_frames or dragon_full_film" > $O/parity.log 2>&1
tail -5 $O/parity.log
timeout 1200 python tools/ab_libraries.py --workloads cornell,dragon,matpreview-rc,matpreview-rd --draws 3 --rounds 2 base=monte-carlo-path-tracing_amd/libmcpt_hip.so cp=$E/cp/libmcpt_hip.so > $O/ab_cp.json 2> $O/ab_cp.err
cat $O/ab_cp.json
timeout 900 python tools/ab_libraries.py --share 8 --workloads cornell,dragon,matpreview-rc --draws 3 --rounds 2 base=monte-carlo-path-tracing_amd/libmcpt_hip.so cp=$E/cp/libmcpt_hip.so > $O/ab_cp_share8.json 2> $O/ab_cp_share8.err
cat $O/ab_cp_share8.json
