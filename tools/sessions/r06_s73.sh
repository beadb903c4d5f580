#!/bin/bash
# round 6, GPU session 73: child-parallel node steps (R5-5) off in dragon's unit, on the steady final kernel
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r06_s73; mkdir -p $O
E=$PWD/monte-carlo-path-tracing_amd/exp
P=$PWD/monte-carlo-path-tracing_amd/libmcpt_hip.so
timeout 900 python tools/ab_libraries.py --workloads dragon --draws 8 --rounds 2 production=$P one_lane_per_item=$E/p2_nochildpar/libmcpt_hip.so > $O/ab.json 2> $O/err.log
python - <<'P'
import json
for line in open('gpurun_out/r06_s73/ab.json'):
    d=json.loads(line)
    for w,r in d.items():
        print(w, 'identical', r['frames_identical'])
        for k,v in r.items():
            if isinstance(v,dict): print('  %-18s %8.2f  (%.2f-%.2f) n=%d %s'%(k, v['median_ms'] or -1, v['min_ms'] or -1, v['max_ms'] or -1, v['n'], v.get('error','')[:200]))
P
