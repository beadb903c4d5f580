#!/bin/bash
# usage: kinfo.sh obj  -> per kernel: vgpr, sgpr, spills, scratch, lds
obj=$1
tmp=$(mktemp -d)
cp $obj $tmp/in.o # (llvm-objcopy without an output file rewrites its input: the build's object keeps its time stamp)
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=$tmp/fb.bin $tmp/in.o
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$tmp/fb.bin --output=$tmp/dev.co --unbundle
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $tmp/dev.co | python3 -c "
import sys,re
txt=sys.stdin.read()
for blk in txt.split('- .agpr_count')[1:]:
    def g(k):
        m=re.search(r'\.'+k+r':\s+(\S+)',blk); return m.group(1) if m else '?'
    name=g('name')
    import subprocess
    print(subprocess.run(['c++filt',name],capture_output=True,text=True).stdout.strip()[:110].replace('mcpt::(anonymous namespace)::',''), 'vgpr',g('vgpr_count'),'sgpr',g('sgpr_count'),'vspill',g('vgpr_spill_count'),'sspill',g('sgpr_spill_count'),'scratch',g('private_segment_fixed_size'),'lds',g('group_segment_fixed_size'))
"
[ -n "$2" ] && cp $tmp/dev.co $2
rm -rf $tmp
