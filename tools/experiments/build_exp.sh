#!/bin/bash
# Experiment build of the library: the production objects with some HIP units recompiled under extra flags.
#   tools/experiments/build_exp.sh <name> "<extra flags>" <unit> [<unit> ...]      (unit = file stem under csrc/hip/)
# -> monte-carlo-path-tracing_amd/exp/<name>/libmcpt_hip.so  (select with MCPT_LIB=<that path>; travels with gpurun)
set -e
NAME=$1; EXTRA=$2; shift 2
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
CSRC=$ROOT/monte-carlo-path-tracing_amd/csrc
OUT=$ROOT/monte-carlo-path-tracing_amd/exp/$NAME
OBJ=$CSRC/build_exp/$NAME
mkdir -p $OUT $OBJ
cd $CSRC
pids=()
for u in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -Wall -Wno-unused-function -I../../include -I. $EXTRA \
      -c hip/$u.hip -o $OBJ/$u.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
objs=()
for o in $(find build -name '*.o'); do
  stem=$(basename $o .o)
  if [ -f $OBJ/$stem.o ]; then objs+=($OBJ/$stem.o); else objs+=($o); fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libmcpt_hip.so "${objs[@]}" -lz
echo "built $OUT/libmcpt_hip.so"
