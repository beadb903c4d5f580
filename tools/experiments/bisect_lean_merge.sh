#!/bin/bash
# Round 6 (EXPERIMENTS R6-1): libraries whose lean LDS-resident MERGED pool-walk kernel (hip/render_variants_lean_pool_merged.hip, only
# Launch<kPM, false, true>) is compiled with LLVM's pass bisection stopped at <limit> — which optimisation turns a kernel that is
# exact at -O1 into one that renders cornell wrong at -O2 / -O3.
#   tools/experiments/bisect_lean_merge.sh <limit> [<limit> ...]    -> monte-carlo-path-tracing_amd/exp/bis_<limit>/libmcpt_hip.so
#   tools/experiments/bisect_lean_merge.sh <name>="<flags>" ...      -> .../exp/bis_<name>/ : the whole pipeline with extra flags instead
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
CSRC=$ROOT/monte-carlo-path-tracing_amd/csrc
cd $CSRC
COMMON="--offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -Wall -Wno-unused-function -I../../include -I. -mllvm -disable-machine-licm"
DISP=$CSRC/build_exp/bis_dispatch
mkdir -p $DISP
# (the dispatcher that chooses the merged form: once)
if [ ! -f $DISP/render_kernel.o ] || [ hip/render_kernel.hip -nt $DISP/render_kernel.o ]; then
  /opt/rocm/bin/hipcc $COMMON -DMCPT_FORCE_LEAN_MERGED=1 -c hip/render_kernel.hip -o $DISP/render_kernel.o
fi
one() {
  L=$1; FLAGS="-mllvm -opt-bisect-limit=$L"
  case "$L" in *=*) FLAGS="${L#*=}"; L="${L%%=*}";; esac
  OBJ=$CSRC/build_exp/bis_$L; OUT=$ROOT/monte-carlo-path-tracing_amd/exp/bis_$L
  mkdir -p $OBJ $OUT
  /opt/rocm/bin/hipcc $COMMON $EXTRA_FLAGS -DMCPT_LEAN_POOL_ONLY_MERGED $FLAGS -c hip/render_variants_lean_pool_merged.hip -o $OBJ/render_variants_lean_pool_merged.o 2> $OBJ/bisect.log
  objs=()
  for o in $(make -n -B ../libmcpt_hip.so 2>/dev/null | grep -- '-shared' | tr ' ' '\n' | grep '^build/.*\.o$'); do
    stem=$(basename $o .o)
    if [ $stem = render_variants_lean_pool_merged ]; then objs+=($OBJ/$stem.o); elif [ $stem = render_kernel ]; then objs+=($DISP/$stem.o); else objs+=($o); fi
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libmcpt_hip.so "${objs[@]}" -lz
  echo "built bis_$L"
}
for L in "$@"; do one "$L" & done
wait
