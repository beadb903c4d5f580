#!/bin/bash
# round 6, GPU session 55: treelet order of the quantised nodes re-measured on the FINAL dragon kernel (R6-5 measured it inside a draw-to-draw
# spread of +-13 %; the frame is steady within 1 % since the path market) — and on matpreview
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r06_s55; mkdir -p $O
H=$PWD/monte-carlo-path-tracing_amd/exp/hooks2/libmcpt_hip.so
timeout 1500 python tools/ab_libraries.py --workloads dragon,matpreview-rc --draws 6 --rounds 2 \
  breadth_first=$H@MCPT_TREELET=0 treelet_8=$H@MCPT_TREELET=8 treelet_32=$H@MCPT_TREELET=32 treelet_128=$H@MCPT_TREELET=128 treelet_512=$H@MCPT_TREELET=512 > $O/ab.json 2> $O/err.log
python - <<'P'
import json
for line in open('gpurun_out/r06_s55/ab.json'):
    d=json.loads(line)
    for w,r in d.items():
        print(w, 'identical', r['frames_identical'])
        for k,v in r.items():
            if isinstance(v,dict): print('  %-14s %8.2f  (%.2f-%.2f) n=%d %s'%(k, v['median_ms'] or -1, v['min_ms'] or -1, v['max_ms'] or -1, v['n'], v.get('error','')[:200]))
P
