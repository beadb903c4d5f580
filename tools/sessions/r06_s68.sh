#!/bin/bash
# round 6, GPU session 68: the general pool-walk units (render_variants_pool, _pool_4: the reference's other scenes) without the SLP vectoriser
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r06_s68; mkdir -p $O
E=$PWD/monte-carlo-path-tracing_amd/exp
P=$PWD/monte-carlo-path-tracing_amd/libmcpt_hip.so
timeout 1500 python tools/ab_libraries.py --workloads other:classroom,other:dining-room,other:box,other:matpreview-rough-plastic,other:matpreview-thin-dielectric --draws 6 --rounds 2 production=$P no_slp=$E/gen_noslp/libmcpt_hip.so > $O/ab.json 2> $O/err.log
python - <<'P'
import json
for line in open('gpurun_out/r06_s68/ab.json'):
    d=json.loads(line)
    for w,r in d.items():
        print(w, 'identical', r['frames_identical'])
        for k,v in r.items():
            if isinstance(v,dict): print('  %-12s %8.2f  (%.2f-%.2f) n=%d %s %s'%(k, v['median_ms'] or -1, v['min_ms'] or -1, v['max_ms'] or -1, v['n'], (v.get('kernel') or '')[:60], v.get('error','')[:200]))
P
