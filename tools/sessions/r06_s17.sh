#!/bin/bash
# round 6, GPU session 17: one division per softmax in the loss kernels (K products instead of K divisions), and the hardware
# reciprocal instead of that division -- three library builds alternated on one box; the loss tests under each
set -u
tag=r06t
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out/$tag
mkdir -p "$out"
cd $repo
lib=advchain_amd/csrc/libadvchain_hip.so
cp $lib /tmp/lib_keep.so
for v in lossrcp losshwrcp; do
  cp scratch/ab/lib_$v.so $lib
  timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_solver_gpu.py -m gpu -q -x -k "loss or bf16 or g6 or golden" > "$out/pytest_$v.log" 2>&1
  echo "$v: $(tail -1 $out/pytest_$v.log)"
done
for v in lossdiv lossrcp losshwrcp lossdiv lossrcp losshwrcp; do
  cp scratch/ab/lib_$v.so $lib
  echo "== build: $v"
  python tools/kernel_bench.py --shape 2d --reps 50 --only "loss" 2>/dev/null | grep -E "loss" | sed "s/^/2d   /"
  python tools/kernel_bench.py --shape 3d --reps 30 --only "loss" 2>/dev/null | grep -E "loss" | sed "s/^/3d   /"
  python tools/kernel_bench.py --shape 3d5 --batch 4 --reps 20 --only "loss" 2>/dev/null | grep -E "loss" | sed "s/^/3d5  /"
  for w in cfg2 cfg2 cfg3 cfg5; do
    python bench.py --workload $w --only-workload --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v $w', d['ms_per_step'], d['value'])"
  done
done > "$out/softmax_rcp_ab.txt" 2>&1
cp /tmp/lib_keep.so $lib
cat "$out/softmax_rcp_ab.txt"
