#!/bin/bash
# round 5, GPU session 1: memory-side counters of the mesh workloads with round 4's library + A/B of dragon's unit under
# the no-machine-LICM / sink flags at 3 and 4 wavefronts per SIMD.
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r05_s1; mkdir -p $O
rocprofv3 -L > $O/counters_list.txt 2>&1
for w in dragon matpreview-rc matpreview-rd; do
  timeout 900 python tools/pmc_groups.py --out $O/pmc_$w.json -- python tools/render_scene.py workload:$w --draws 1 > $O/pmc_$w.log 2>&1
done
E=monte-carlo-path-tracing_amd/exp
timeout 900 python tools/ab_libraries.py --workloads dragon --draws 5 --rounds 3 base=monte-carlo-path-tracing_amd/libmcpt_hip.so flags3=$E/d_flags3/libmcpt_hip.so flags4=$E/d_flags4/libmcpt_hip.so > $O/ab_dragon_flags.json 2> $O/ab_dragon_flags.err
tail -c 1500 $O/ab_dragon_flags.json
for w in dragon matpreview-rc matpreview-rd; do tail -c 600 $O/pmc_$w.log; done
