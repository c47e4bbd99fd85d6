#!/bin/bash
# One source recompiled with extra flags, linked with the other objects of the last build into a variant library:
#   tools/ab/build_variant.sh <source.hip> <out.so> [-DKNOB=1 ...]        (then: tools/ab/lib_ab.sh a.so b.so -- <command>)
set -e
src=$1; out=$2; shift 2
here=$(cd "$(dirname "$0")/../.." && pwd)
csrc=$here/advchain_amd/csrc
tl=$(python -c "import torch, os; print(os.path.join(os.path.dirname(torch.__file__), 'lib'))")
extra=""
case $(basename $src) in adjoint_march.hip|adjoint_gather.hip|adjoint_fused2d.hip|fields.hip) extra="-fno-slp-vectorize";; esac
obj=/tmp/variant_$(basename $src).$$.o
/opt/rocm/bin/hipcc -c --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -I$here/include -I$csrc $extra "$@" $csrc/$(basename $src) -o $obj
objs=$(ls $csrc/build/*.o | grep -v "/$(basename $src).o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $obj -o $out -no-hip-rt -L$tl -lamdhip64 -Wl,-rpath,$tl
rm -f $obj
echo "built $out"
