#!/bin/bash
# round 6, GPU session 67: scheduler strategies / -O2 on dragon's unit, on top of its new flags (no SLP, no machine LICM, sinking)
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r06_s67; mkdir -p $O
E=$PWD/monte-carlo-path-tracing_amd/exp
P=$PWD/monte-carlo-path-tracing_amd/libmcpt_hip.so
timeout 1500 python tools/ab_libraries.py --workloads dragon --draws 8 --rounds 2 production=$P max_memory_clause=$E/p2_memclause/libmcpt_hip.so iterative_ilp=$E/p2_iter_ilp/libmcpt_hip.so \
   iterative_maxocc=$E/p2_iter_maxocc/libmcpt_hip.so max_ilp=$E/p2_maxilp2/libmcpt_hip.so no_sink_aa_in_codegen=$E/p2_nosink_aa/libmcpt_hip.so O2=$E/p2_O2/libmcpt_hip.so > $O/ab.json 2> $O/err.log
python - <<'P'
import json
for line in open('gpurun_out/r06_s67/ab.json'):
    d=json.loads(line)
    for w,r in d.items():
        print(w, 'identical', r['frames_identical'])
        for k,v in r.items():
            if isinstance(v,dict): print('  %-22s %8.2f  (%.2f-%.2f) n=%d %s'%(k, v['median_ms'] or -1, v['min_ms'] or -1, v['max_ms'] or -1, v['n'], v.get('error','')[:200]))
P
