#!/bin/bash
# round 6, GPU session 18: node items that name PAIRS of quantised records (-DMCPT_POOL_PAIRS=1) against the 4-wide records: dragon, matpreview; dragon's slowest tiles alone
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r06_s18; mkdir -p $O
L=monte-carlo-path-tracing_amd
timeout 900 python tools/ab_libraries.py --workloads dragon --draws 12 --rounds 2 four_wide=$L/libmcpt_hip.so pairs=$L/exp/pairs/libmcpt_hip.so > $O/ab_dragon.jsonl 2> $O/ab_dragon.err
cut -c1-400 $O/ab_dragon.jsonl; tail -3 $O/ab_dragon.err
MCPT_TILES="81,63;93,60" timeout 300 python tools/experiments/dragon_slowest_tile.py dragon > $O/tiles_four_wide.jsonl 2> $O/tiles_four_wide.err; cat $O/tiles_four_wide.jsonl
MCPT_LIB=$PWD/$L/exp/pairs/libmcpt_hip.so MCPT_TILES="81,63;93,60" timeout 300 python tools/experiments/dragon_slowest_tile.py dragon > $O/tiles_pairs.jsonl 2> $O/tiles_pairs.err; cat $O/tiles_pairs.jsonl
timeout 1200 python tools/ab_libraries.py --workloads matpreview-rc,matpreview-rd --draws 5 --rounds 2 four_wide=$L/libmcpt_hip.so pairs=$L/exp/pairs/libmcpt_hip.so > $O/ab_matpreview.jsonl 2> $O/ab_matpreview.err
cut -c1-300 $O/ab_matpreview.jsonl
