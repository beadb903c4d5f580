#!/bin/bash
# round 6, GPU session 53: instruction-issue priority by a wavefront's lanes-per-path level (s_setprio(level): the frame's longest chains
# first) in matpreview's units, at three values of kappa (hooks builds) — EXPERIMENTS R6-19
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r06_s53; mkdir -p $O
E=$PWD/monte-carlo-path-tracing_amd/exp
H=$E/hooks/libmcpt_hip.so; A=$E/lprio1/libmcpt_hip.so; B=$E/lprio2/libmcpt_hip.so
timeout 1500 python tools/ab_libraries.py --workloads matpreview-rc,matpreview-rd --draws 4 --rounds 2 \
  production=$PWD/monte-carlo-path-tracing_amd/libmcpt_hip.so \
  hooks_k0.6=$H@MCPT_LEVEL_KAPPA=0.6 prio_by_level_k0.6=$A@MCPT_LEVEL_KAPPA=0.6 prio_any_level_k0.6=$B@MCPT_LEVEL_KAPPA=0.6 \
  hooks_k0.8=$H@MCPT_LEVEL_KAPPA=0.8 prio_by_level_k0.8=$A@MCPT_LEVEL_KAPPA=0.8 prio_any_level_k0.8=$B@MCPT_LEVEL_KAPPA=0.8 \
  hooks_k1.0=$H@MCPT_LEVEL_KAPPA=1.0 prio_by_level_k1.0=$A@MCPT_LEVEL_KAPPA=1.0 \
  hooks_k0.45=$H@MCPT_LEVEL_KAPPA=0.45 prio_by_level_k0.45=$A@MCPT_LEVEL_KAPPA=0.45 > $O/ab.json 2> $O/err.log
python - <<'P'
import json
for line in open('gpurun_out/r06_s53/ab.json'):
    d=json.loads(line)
    for w,r in d.items():
        print(w, 'identical', r['frames_identical'])
        for k,v in r.items():
            if isinstance(v,dict): print('  %-24s %8.2f  (%.2f-%.2f) n=%d %s'%(k, v['median_ms'] or -1, v['min_ms'] or -1, v['max_ms'] or -1, v['n'], v.get('error','')[:200]))
P
