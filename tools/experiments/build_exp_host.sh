#!/bin/bash
# Experiment build of the library (round 6): the production objects with some HOST units (csrc/capi.cpp, csrc/host/*.cpp) recompiled
# under extra flags — e.g. -DMCPT_MEASUREMENT_HOOKS=1 (the environment switches of csrc/host/measurement_env.hpp), -DMCPT_WIDE_TREELET=32.
#   tools/experiments/build_exp_host.sh <name> "<extra flags>" <unit> ...      (unit = capi, or a file stem under csrc/host/)
# -> monte-carlo-path-tracing_amd/exp/<name>/libmcpt_hip.so  (select with MCPT_LIB=<that path>; travels with gpurun)
set -e
NAME=$1; EXTRA=$2; shift 2
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
CSRC=$ROOT/monte-carlo-path-tracing_amd/csrc
OUT=$ROOT/monte-carlo-path-tracing_amd/exp/$NAME
OBJ=$CSRC/build_exp/$NAME
mkdir -p $OUT $OBJ
cd $CSRC
for u in "$@"; do
  src=host/$u.cpp; [ $u = capi ] && src=capi.cpp
  /opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -Wall -Wno-unused-function -I../../include -I. $EXTRA -c $src -o $OBJ/$u.o &
done
wait
objs=()
for o in $(make -n -B ../libmcpt_hip.so 2>/dev/null | grep -- '-shared' | tr ' ' '\n' | grep '^build/.*\.o$'); do
  stem=$(basename $o .o)
  if [ -f $OBJ/$stem.o ]; then objs+=($OBJ/$stem.o); else objs+=($o); fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libmcpt_hip.so "${objs[@]}" -lz
echo "built $OUT/libmcpt_hip.so"
