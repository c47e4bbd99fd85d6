#!/bin/bash
# round 6, GPU session 6: one-instruction fixed-point rounding in the LDS-integer scatters + the trimmed ring-slot arithmetic of
# the wide march scatter (A/B of two library builds on one box), parity first
set -u
tag=r06g
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out/$tag
mkdir -p "$out"
cd $repo
timeout 1500 python -m pytest tests -m gpu -q -x > "$out/pytest_gpu.log" 2>&1
tail -4 "$out/pytest_gpu.log"
lib=advchain_amd/csrc/libadvchain_hip.so
cp $lib /tmp/lib_keep.so
for v in norpi rpi norpi rpi; do
  cp scratch/ab/lib_$v.so $lib
  echo "== build: $v"
  bash tools/profile_ns_pair.sh ${tag}_$v > /dev/null 2>&1
  cat gpurun_out/${tag}_$v/ns_pair_summary.txt | grep -E "^level|scatter"
  python tools/kernel_bench.py --shape 3d --batch 8 --reps 30 --only "march)" 2>/dev/null | grep -E "march" | sed "s/^/N=8  /"
  python tools/kernel_bench.py --shape 3d5 --reps 20 --only "march)" 2>/dev/null | grep -E "compose_self bwd" | sed "s/^/3d5  /"
  python tools/kernel_bench.py --shape 2d --reps 30 --only "halo=-" 2>/dev/null | grep -E "compose_self bwd (3.5|6.0|12.0) px halo=-|C=4 (3.5|12.0) px halo=-" | sed "s/^/2d   /"
  for w in cfg2 cfg3 cfg4 cfg5; do
    python bench.py --workload $w --only-workload --steps 8 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v $w', d['ms_per_step'], d['value'], d['roofline']['frac'] if d['roofline'] else None)"
  done
done > "$out/rpi_ab.txt" 2>&1
cp /tmp/lib_keep.so $lib
cat "$out/rpi_ab.txt"
