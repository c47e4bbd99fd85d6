#!/bin/bash
# rocprofv3 kernel trace of the north-star pair ALONE (tools/north_star_pair.py) at the two displacement levels bench.py
# reports, so that fwd_us / bwd_us of the bench line can be recomputed from a committed CSV:
#   tools/profile_ns_pair.sh <tag>   ->  gpurun_out/<tag>/ns_{init_field,after_cfg3_ascent}_kernel_stats.csv + ns_pair_summary.txt
set -u
tag=${1:-prof}
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
python $repo/tools/north_star_pair.py --make-fields /tmp/ns_fields.pt > "$out/ns_make_fields.log" 2>&1
for level in init_field after_cfg3_ascent; do
  name=ns_$level
  rm -rf /tmp/rp_$name
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$name -o $name -- \
      python $repo/tools/north_star_pair.py --fields /tmp/ns_fields.pt --level $level --reps 50 > "$out/${name}_under_rocprof.log" 2>&1
  cp "$(find /tmp/rp_$name -name "${name}_kernel_stats.csv" | head -1)" "$out/${name}_kernel_stats.csv"
done
python $repo/tools/ns_pair_summary.py "$out" > "$out/ns_pair_summary.txt" 2>&1
cat "$out/ns_pair_summary.txt"
