#!/bin/bash
# round 6, GPU session 70: matpreview's unit without SLP — with machine LICM back / without the sinking (two more cells of R6-21's table)
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r06_s70; mkdir -p $O
E=$PWD/monte-carlo-path-tracing_amd/exp
P=$PWD/monte-carlo-path-tracing_amd/libmcpt_hip.so
timeout 1500 python tools/ab_libraries.py --workloads matpreview-rc,matpreview-rd --draws 5 --rounds 2 production=$P no_slp_only=$E/p3_noslp_only/libmcpt_hip.so no_slp_no_mlicm=$E/p3_noslp_nomlicm/libmcpt_hip.so > $O/ab.json 2> $O/err.log
python - <<'P'
import json
for line in open('gpurun_out/r06_s70/ab.json'):
    d=json.loads(line)
    for w,r in d.items():
        print(w, 'identical', r['frames_identical'])
        for k,v in r.items():
            if isinstance(v,dict): print('  %-18s %8.2f  (%.2f-%.2f) n=%d %s'%(k, v['median_ms'] or -1, v['min_ms'] or -1, v['max_ms'] or -1, v['n'], v.get('error','')[:200]))
P
