#!/bin/bash
# round 6, GPU session 15: branch-light deposits in the affine grad_in scatter + the select form of lin_coord (A/B of two library
# builds on one box), parity first
set -u
tag=r06s
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out/$tag
mkdir -p "$out"
cd $repo
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_ride_gpu.py tests/test_fused2d_gpu.py tests/test_solver_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x > "$out/pytest_subset.log" 2>&1
tail -4 "$out/pytest_subset.log"
lib=advchain_amd/csrc/libadvchain_hip.so
cp $lib /tmp/lib_keep.so
for v in ginbase ginflat ginbase ginflat; do
  cp scratch/ab/lib_$v.so $lib
  echo "== build: $v"
  python tools/kernel_bench.py --shape 2d --reps 30 --only "affine" 2>/dev/null | grep -E "affine" | sed "s/^/2d   /"
  python tools/kernel_bench.py --shape 3d --reps 20 --only "affine" 2>/dev/null | grep -E "affine" | sed "s/^/3d   /"
  for w in cfg2 cfg2 cfg3 cfg4; do
    python bench.py --workload $w --only-workload --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v $w', d['ms_per_step'], d['value'])"
  done
done > "$out/gin_flat_ab.txt" 2>&1
cp /tmp/lib_keep.so $lib
cat "$out/gin_flat_ab.txt"
