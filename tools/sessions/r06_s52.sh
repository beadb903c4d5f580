#!/bin/bash
# round 6, GPU session 52: instruction-issue priorities (s_setprio) of the wavefronts that share a SIMD on cornell — static by cost
# quarter (expensive high / cheap high) and paced against the launch's mean progress (EXPERIMENTS R6-18); PRIM_AT=48 beside them
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r06_s52; mkdir -p $O
E=$PWD/monte-carlo-path-tracing_amd/exp
timeout 900 python tools/ab_libraries.py --workloads cornell --draws 8 --rounds 2 production=$PWD/monte-carlo-path-tracing_amd/libmcpt_hip.so \
  static_expensive_high=$E/pace1/libmcpt_hip.so static_cheap_high=$E/pace2/libmcpt_hip.so paced_8=$E/pace3/libmcpt_hip.so paced_4=$E/pace3_s4/libmcpt_hip.so \
  paced_16=$E/pace3_s16/libmcpt_hip.so paced_4_band2=$E/pace3_b2/libmcpt_hip.so primat48=$E/primat48/libmcpt_hip.so > $O/ab.json 2> $O/err.log
tail -c 3000 $O/ab.json
