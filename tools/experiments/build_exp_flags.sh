#!/bin/bash
# Experiment build of the library with ONE HIP unit compiled under exactly the given back-end flags (instead of the ones csrc/Makefile gives it):
#   tools/experiments/build_exp_flags.sh <name> <unit> "<flags>"   -> monte-carlo-path-tracing_amd/exp/<name>/libmcpt_hip.so
set -e
NAME=$1; U=$2; FLAGS=$3
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
CSRC=$ROOT/monte-carlo-path-tracing_amd/csrc
OUT=$ROOT/monte-carlo-path-tracing_amd/exp/$NAME; OBJ=$CSRC/build_exp/$NAME
mkdir -p $OUT $OBJ
cd $CSRC
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -Wall -Wno-unused-function -I../../include -I. $FLAGS -c hip/$U.hip -o $OBJ/$U.o
objs=()
for o in $(make -n -B ../libmcpt_hip.so 2>/dev/null | grep -- '-shared' | tr ' ' '\n' | grep '^build/.*\.o$'); do
  stem=$(basename $o .o)
  if [ -f $OBJ/$stem.o ]; then objs+=($OBJ/$stem.o); else objs+=($o); fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libmcpt_hip.so "${objs[@]}" -lz
echo "built $OUT/libmcpt_hip.so"
