#!/bin/bash
# usage: variant2.sh TAG "flags" "extra for pool TUs" "extra for sorted TU"
tag=$1; fl=$2; xp=$3; xs=$4; src=/root/repo/monte-carlo-path-tracing_amd/csrc; out=/tmp/exp/v_$tag; mkdir -p $out
cd $src
pids=()
for tu in render_kernel sorted_kernel render_variants_pool_2 render_variants_pool_3; do
  x=""; case $tu in render_variants_pool_*) x=$xp;; sorted_kernel) x=$xs;; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -Wno-unused-function -I../../include -I. $fl $x -c hip/$tu.hip -o $out/$tu.o > $out/$tu.log 2>&1 &
  pids+=($!)
done
wait "${pids[@]}"
objs=$(find build -name '*.o' | grep -v -E "hip/(render_kernel|sorted_kernel|render_variants_pool_2|render_variants_pool_3)\.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/scratch/libs/libmcpt_hip_$tag.so $objs $out/*.o -lz && echo built $tag
for tu in render_kernel sorted_kernel render_variants_pool_2 render_variants_pool_3; do bash /root/repo/tools/kernel_resources.sh $out/$tu.o | grep -o "[a-z_]*kernel<[0-9]*u[^>]*>\|vgpr.*" | paste - - ; done
