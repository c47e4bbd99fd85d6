#!/bin/bash
# round 6, GPU session 16: the bf16 STORAGE experiment of the 2D K = 4 fused loss (fp32 against bf16 storage, same values)
set -u
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out/r06bf
mkdir -p "$out"
cd $repo
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "loss or bf16" > "$out/pytest_loss.log" 2>&1
tail -4 "$out/pytest_loss.log"
for rep in 1 2 3; do
  python tools/kernel_bench.py --shape 2d --reps 50 --only "fused loss" 2>/dev/null | grep "fused loss"
done | tee "$out/bf16_storage.txt"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o bf16 -- python $repo/tools/kernel_bench.py --shape 2d --reps 50 --only "fused loss" > /dev/null 2>&1
f=$(find $out/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $out/bf16_kernel_stats.csv && grep "k_loss_fused" $out/bf16_kernel_stats.csv | cut -c1-260
rm -rf $out/prof
