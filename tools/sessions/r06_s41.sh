#!/bin/bash
# round 6, GPU session 41: the phase clock on the final kernels of matpreview and dragon (where the lanes are now that a sample's successor starts in the same step)
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r06_s41; mkdir -p $O
export MCPT_LIB=$PWD/monte-carlo-path-tracing_amd/exp/phase/libmcpt_hip.so
for w in matpreview-rc matpreview-rd dragon; do timeout 400 python tools/experiments/phase_clock.py $w --out $O/phase_clock_$w.json > $O/phase_$w.log 2>&1; tail -c 300 $O/phase_$w.log; done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_s41/phase_clock_*.json')):
    d=json.load(open(f)); print(f.split('/')[-1], round(d.get('kernel_ms',0),1), d.get('lane_bound'))
    for p in d['phases']: print('   ', p['phase'][:52].ljust(52), p['share'], p.get('lanes_at_mark'))
P
