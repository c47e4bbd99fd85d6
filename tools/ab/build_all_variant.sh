#!/bin/bash
# The whole library recompiled with extra -D flags into a variant .so (for A/B on one box with tools/ab/lib_ab.sh):
#   tools/ab/build_all_variant.sh <out.so> -DADVCHAIN_FIX_RPI=0 ...
set -e
out=$1; shift
here=$(cd "$(dirname "$0")/../.." && pwd)
csrc=$here/advchain_amd/csrc
tl=$(python -c "import torch, os; print(os.path.join(os.path.dirname(torch.__file__), 'lib'))")
tmp=$(mktemp -d)
pids=()
for src in $csrc/*.hip $csrc/*.cpp; do
  extra=""
  case $(basename $src) in adjoint_march.hip|adjoint_gather.hip|adjoint_fused2d.hip|fields.hip) extra="-fno-slp-vectorize";; esac
  /opt/rocm/bin/hipcc -c --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -I$here/include -I$csrc $extra "$@" $src -o $tmp/$(basename $src).o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $tmp/*.o -o $out -no-hip-rt -L$tl -lamdhip64 -Wl,-rpath,$tl
rm -rf $tmp
echo "built $out"
