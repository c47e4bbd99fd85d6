#!/bin/bash
# round 6, GPU session 10: branch-light deposits in the window scatters (A/B of two library builds on one box), parity first
set -u
tag=r06n
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out/$tag
mkdir -p "$out"
cd $repo
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_deterministic_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x > "$out/pytest_subset.log" 2>&1
tail -4 "$out/pytest_subset.log"
lib=advchain_amd/csrc/libadvchain_hip.so
cp $lib /tmp/lib_keep.so
for v in winbranchy winflat winbranchy winflat; do
  cp scratch/ab/lib_$v.so $lib
  echo "== build: $v"
  python tools/kernel_bench.py --shape 2d --batch 64 --reps 30 --only "halo=16" 2>/dev/null | grep -E "halo=16" | sed "s/^/2d   /"
  python tools/kernel_bench.py --shape 3d --batch 8 --reps 20 --only "(window)" 2>/dev/null | grep -E "window" | sed "s/^/N=8  /"
  python tools/kernel_bench.py --shape 3d5 --reps 10 --only "(window)" 2>/dev/null | grep -E "window" | sed "s/^/3d5  /"
  for w in cfg2 cfg2 cfg5 cfg3; do
    python bench.py --workload $w --only-workload --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v $w', d['ms_per_step'], d['value'])"
  done
done > "$out/window_flat_ab.txt" 2>&1
cp /tmp/lib_keep.so $lib
cat "$out/window_flat_ab.txt"
