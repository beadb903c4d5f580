#!/bin/bash
# round 6, GPU session 71: sibling alignment of the quantised node records (MCPT_WIDE_ALIGN, R5: "dragon +-0 inside its spread" — a spread of +-13 % then)
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r06_s71; mkdir -p $O
H=$PWD/monte-carlo-path-tracing_amd/exp/hooks2/libmcpt_hip.so
timeout 900 python tools/ab_libraries.py --workloads dragon --draws 8 --rounds 2 as_stored=$H@MCPT_WIDE_ALIGN=0 siblings_aligned=$H@MCPT_WIDE_ALIGN=1 > $O/ab.json 2> $O/err.log
python - <<'P'
import json
for line in open('gpurun_out/r06_s71/ab.json'):
    d=json.loads(line)
    for w,r in d.items():
        print(w, 'identical', r['frames_identical'])
        for k,v in r.items():
            if isinstance(v,dict): print('  %-18s %8.2f  (%.2f-%.2f) n=%d %s'%(k, v['median_ms'] or -1, v['min_ms'] or -1, v['max_ms'] or -1, v['n'], v.get('error','')[:200]))
P
