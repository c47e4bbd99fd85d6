#!/bin/bash
# round 6, GPU session 7: branch-light deposits in the 2D whole-row scatter (A/B of two library builds on one box), parity first
set -u
tag=r06h
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out/$tag
mkdir -p "$out"
cd $repo
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_fused2d_gpu.py tests/test_solver_gpu.py tests/test_graph_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x > "$out/pytest_subset.log" 2>&1
tail -4 "$out/pytest_subset.log"
lib=advchain_amd/csrc/libadvchain_hip.so
cp $lib /tmp/lib_keep.so
for v in branchy flat branchy flat; do
  cp scratch/ab/lib_$v.so $lib
  echo "== build: $v"
  python tools/kernel_bench.py --shape 2d --reps 40 --only "halo=-" 2>/dev/null | grep -E "halo=-" | sed "s/^/2d   /"
  python tools/kernel_bench.py --shape 2d --reps 20 --only "expo_chain bwd" 2>/dev/null | grep -E "expo_chain" | sed "s/^/2d   /"
  for w in cfg2 cfg2 cfg1; do
    python bench.py --workload $w --only-workload --steps 20 --warmup 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v $w', d['ms_per_step'], d['value'])"
  done
done > "$out/rows2d_flat_ab.txt" 2>&1
cp /tmp/lib_keep.so $lib
cat "$out/rows2d_flat_ab.txt"
