#!/bin/bash
# Experiment build of the library (round 5): the production objects with some HIP units recompiled under extra flags, each unit
# keeping the per-unit back-end flags csrc/Makefile gives it (asked from make itself) unless it is listed as unit=noflags.
#   tools/experiments/build_exp2.sh <name> "<extra flags for every listed unit>" <unit>[+"more flags"] ...
# -> monte-carlo-path-tracing_amd/exp/<name>/libmcpt_hip.so  (select with MCPT_LIB=<that path>; travels with gpurun)
set -e
NAME=$1; EXTRA=$2; shift 2
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
CSRC=$ROOT/monte-carlo-path-tracing_amd/csrc
OUT=$ROOT/monte-carlo-path-tracing_amd/exp/$NAME
OBJ=$CSRC/build_exp/$NAME
mkdir -p $OUT $OBJ
cd $CSRC
JOBS=${JOBS:-8}
list=()
for spec in "$@"; do
  u=${spec%%+*}; more=""; [ "$spec" != "$u" ] && more=${spec#*+}
  # the unit's own flags, as the Makefile would pass them
  uf=$(make -n -B build/hip/$u.o 2>/dev/null | grep -- "-c hip/$u.hip" | sed -e 's/.*-I\. *//' -e 's/ *-MMD.*//')
  list+=("$u|$uf $EXTRA $more")
done
printf '%s\n' "${list[@]}" | xargs -P $JOBS -I{} bash -c 'spec="{}"; u=${spec%%|*}; f=${spec#*|}; /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -Wall -Wno-unused-function -I../../include -I. $f -c hip/$u.hip -o '$OBJ'/$u.o || echo "FAILED $u"'
objs=()
# (the objects of the default library: make's own list, not everything an EXPERIMENTAL=1 build may have left under build/)
for o in $(make -n -B ../libmcpt_hip.so 2>/dev/null | grep -- '-shared' | tr ' ' '\n' | grep '^build/.*\.o$'); do
  stem=$(basename $o .o)
  if [ -f $OBJ/$stem.o ]; then objs+=($OBJ/$stem.o); else objs+=($o); fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libmcpt_hip.so "${objs[@]}" -lz
echo "built $OUT/libmcpt_hip.so"
