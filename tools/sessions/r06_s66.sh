#!/bin/bash
# round 6, GPU session 66: back-end flags of the camera-ray pre-pass unit (primary_kernel: compiler defaults; 9 % of dragon's frame)
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r06_s66; mkdir -p $O
E=$PWD/monte-carlo-path-tracing_amd/exp
P=$PWD/monte-carlo-path-tracing_amd/libmcpt_hip.so
timeout 1500 python tools/ab_libraries.py --workloads dragon,matpreview-rc --draws 6 --rounds 2 production=$P no_mlicm=$E/pk_nomlicm/libmcpt_hip.so no_mlicm_sink=$E/pk_nomlicm_sink/libmcpt_hip.so \
   no_slp=$E/pk_noslp/libmcpt_hip.so max_ilp=$E/pk_maxilp/libmcpt_hip.so > $O/ab_pk.json 2> $O/err.log
python - <<'P'
import json
for line in open('gpurun_out/r06_s66/ab_pk.json'):
    d=json.loads(line)
    for w,r in d.items():
        print(w, 'identical', r['frames_identical'])
        for k,v in r.items():
            if isinstance(v,dict): print('  %-16s %8.2f  (%.2f-%.2f) n=%d %s'%(k, v['median_ms'] or -1, v['min_ms'] or -1, v['max_ms'] or -1, v['n'], v.get('error','')[:200]))
P
