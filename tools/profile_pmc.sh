#!/bin/bash
# HBM-side traffic counters (separate passes: FETCH_SIZE and WRITE_SIZE do not fit one pass) for bench.py and for the
# per-kernel micro-benchmark.   tools/profile_pmc.sh <tag>  ->  gpurun_out/<tag>/pmc_{fetch,write}_size_{cfg2,cfg3,kb3d}.csv
set -u
tag=${1:-pmc}
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
run() {  # counter, name, command...
  local ctr=$1 name=$2; shift 2
  local lc=$(echo $ctr | tr 'A-Z' 'a-z')
  rm -rf /tmp/pmc_${lc}_$name
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_${lc}_$name -o $name -- "$@" > /tmp/pmc_${lc}_$name.log 2>&1
  python $repo/tools/pmc_summary.py /tmp/pmc_${lc}_$name "$out/pmc_${lc}_$name.csv"
}
for ctr in FETCH_SIZE WRITE_SIZE; do
  run $ctr cfg2 python $repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline
  run $ctr cfg3 python $repo/bench.py --workload cfg3 --steps 2 --warmup 1 --no-cpu-baseline
  run $ctr kb3d python $repo/tools/kernel_bench.py --shape 3d --reps 3
done
ls -la "$out"
