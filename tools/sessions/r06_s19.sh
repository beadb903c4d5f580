#!/bin/bash
# round 6, GPU session 19: transposed hand-out (a wavefront's lanes take pixels of 64 different tiles) on the kernels outside LDS
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r06_s19; mkdir -p $O
timeout 1500 python tools/experiments/pixel_order_ab.py dragon,matpreview-rc,matpreview-rd 10 > $O/pixel_order.jsonl 2> $O/pixel_order.err
cat $O/pixel_order.jsonl; tail -3 $O/pixel_order.err
