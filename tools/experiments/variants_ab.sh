#!/bin/bash
cd /root/repo
for w in dragon matpreview-rc matpreview-rd; do
  for sub in 1 0; do
    echo "== $w MCPT_POOL_SUBSETS=$sub"
    MCPT_POOL_SUBSETS=$sub timeout 300 python tools/render_scene.py workload:$w --draws 4 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(round(d['kernel_ms'],2),'ms',round(d['msamples_per_s'],1),'Ms/s',d['kernel'][:100])"
  done
done
