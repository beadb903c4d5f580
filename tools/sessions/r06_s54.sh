#!/bin/bash
# round 6, GPU session 54: phase clocks of the FINAL cornell and volumetric-caustic kernels; the primitive phase at 96 / 48 waiting slots on
# all four pool-walk workloads (R4-9's constant on the final kernels)
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r06_s54; mkdir -p $O
E=$PWD/monte-carlo-path-tracing_amd/exp
for w in cornell volumetric; do
  MCPT_LIB=$E/phase/libmcpt_hip.so timeout 300 python tools/experiments/phase_clock.py $w --out $O/phase_clock_$w.json > /dev/null 2>> $O/err.log
done
timeout 1200 python tools/ab_libraries.py --workloads cornell,dragon,matpreview-rc,matpreview-rd --draws 5 --rounds 2 production=$PWD/monte-carlo-path-tracing_amd/libmcpt_hip.so \
  primat96=$E/primat96/libmcpt_hip.so primat48=$E/primat48/libmcpt_hip.so > $O/ab.json 2>> $O/err.log
python - <<'P'
import json
for w in ['cornell','volumetric']:
    try:
        d=json.load(open('gpurun_out/r06_s54/phase_clock_%s.json'%w)); print(w, d['kernel_ms'], d['lane_bound'])
        for p in d['phases']: print('   %-58s %.3f lanes %.1f cyc %.0f'%(p['phase'][:58],p['share'],p['lanes_at_mark'],p['cycles_per_mark']))
    except Exception as e: print(w, 'failed', e)
for line in open('gpurun_out/r06_s54/ab.json'):
    d=json.loads(line)
    for w,r in d.items():
        print(w, 'identical', r['frames_identical'])
        for k,v in r.items():
            if isinstance(v,dict): print('  %-14s %8.2f  (%.2f-%.2f) n=%d %s'%(k, v['median_ms'] or -1, v['min_ms'] or -1, v['max_ms'] or -1, v['n'], v.get('error','')[:200]))
P
tail -5 $O/err.log
