#!/bin/bash
# round 6, GPU session 63: the back-end flags of the mesh units (csrc/Makefile, chosen in round 4 / round 6) re-measured on the final kernels
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r06_s63; mkdir -p $O
E=$PWD/monte-carlo-path-tracing_amd/exp
P=$PWD/monte-carlo-path-tracing_amd/libmcpt_hip.so
timeout 1500 python tools/ab_libraries.py --workloads matpreview-rc,matpreview-rd --draws 5 --rounds 2 production=$P defaults=$E/p3_defaults/libmcpt_hip.so no_mlicm=$E/p3_nomlicm/libmcpt_hip.so \
   production_no_slp=$E/p3_prod_noslp/libmcpt_hip.so production_max_ilp=$E/p3_maxilp/libmcpt_hip.so > $O/ab_p3.json 2> $O/err.log
timeout 900 python tools/ab_libraries.py --workloads dragon --draws 8 --rounds 2 production=$P no_slp_no_mlicm=$E/p2_noslp_nomlicm/libmcpt_hip.so no_slp_no_mlicm_sink=$E/p2_noslp_nomlicm_sink/libmcpt_hip.so \
   no_slp_max_ilp=$E/p2_maxilp/libmcpt_hip.so > $O/ab_p2.json 2>> $O/err.log
python - <<'P'
import json
for f in ['ab_p3.json','ab_p2.json']:
  for line in open('gpurun_out/r06_s63/'+f):
    d=json.loads(line)
    for w,r in d.items():
        print(w, 'identical', r['frames_identical'])
        for k,v in r.items():
            if isinstance(v,dict): print('  %-22s %8.2f  (%.2f-%.2f) n=%d %s'%(k, v['median_ms'] or -1, v['min_ms'] or -1, v['max_ms'] or -1, v['n'], v.get('error','')[:200]))
P
