#!/bin/bash
# HBM-side traffic counters (separate passes: FETCH_SIZE and WRITE_SIZE do not fit one pass) for bench.py and for the
# (bench.py runs the ordinary launch-by-launch path here: counter passes serialise every kernel, the kernels are the same)
# per-kernel micro-benchmark.   tools/profile_pmc.sh <tag>  ->  gpurun_out/<tag>/pmc_{fetch,write}_size_{cfg2,cfg3,cfg4,cfg5,kb3d}.csv
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
  timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_${lc}_$name -o $name -- "$@" > /tmp/pmc_${lc}_$name.log 2>&1
  python $repo/tools/pmc_summary.py /tmp/pmc_${lc}_$name "$out/pmc_${lc}_$name.csv"
}
python $repo/tools/north_star_pair.py --make-fields /tmp/ns_fields.pt > /tmp/ns_make.log 2>&1     # not profiled
for ctr in FETCH_SIZE WRITE_SIZE; do
  run $ctr cfg2 python $repo/bench.py --steps 2 --warmup 1 --only-workload --no-graph
  run $ctr cfg3 python $repo/bench.py --workload cfg3 --steps 2 --warmup 1 --only-workload --no-graph
  run $ctr cfg4 python $repo/bench.py --workload cfg4 --steps 2 --warmup 1 --only-workload --no-graph
  run $ctr cfg5 python $repo/bench.py --workload cfg5 --steps 1 --warmup 1 --only-workload --no-graph
  run $ctr kb3d python $repo/tools/kernel_bench.py --shape 3d --reps 3
  run $ctr ns_init_field python $repo/tools/north_star_pair.py --fields /tmp/ns_fields.pt --level init_field --reps 5
  run $ctr ns_after_cfg3_ascent python $repo/tools/north_star_pair.py --fields /tmp/ns_fields.pt --level after_cfg3_ascent --reps 5
done
ls -la "$out"
