#!/bin/bash
# Collect the rocprofv3 kernel-trace summaries that profiles/ holds (run on the GPU box through gpurun).
#   tools/profile_bench.sh <tag>      -> gpurun_out/<tag>/{cfg2,cfg3,kb3d}_kernel_stats.csv + logs
set -u
tag=${1:-prof}
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
run() {  # name, command...
  local name=$1; shift
  # an untraced run first: MIOpen's find pass for the user model's convolutions lands in the user find-db of this box and is
  # not repeated (and not traced) by the profiled run -- `GPU busy` then compares with the driver's ms_per_step
  "$@" > /dev/null 2>&1
  rm -rf /tmp/rp_$name
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$name -o $name -- "$@" > "$out/${name}_under_rocprof.log" 2>&1
  cp "$(find /tmp/rp_$name -name "${name}_kernel_stats.csv" | head -1)" "$out/${name}_kernel_stats.csv"
}
run cfg2 python $repo/bench.py --steps 10 --warmup 3 --only-workload
run cfg3 python $repo/bench.py --workload cfg3 --steps 5 --warmup 2 --only-workload
run cfg5 python $repo/bench.py --workload cfg5 --steps 3 --warmup 1 --only-workload
run kb3d python $repo/tools/kernel_bench.py --shape 3d
# per-call figures: the number of adversarial_training calls the profiled command made is in ITS OWN json line
# ("solver_calls_in_process"), not assumed here (round 4 divided cfg-5's 5 calls by 4)
for w in cfg2 cfg3 cfg5; do
  calls=$(python - "$out/${w}_under_rocprof.log" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        print(json.loads(line)["solver_calls_in_process"])
        break
PY
)
  python $repo/tools/stats_per_call.py "$out/${w}_kernel_stats.csv" "$calls" 30
done
