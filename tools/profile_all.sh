#!/bin/bash
# Everything profiles/rNN holds, in one gpurun call:  tools/profile_all.sh <tag>  ->  gpurun_out/<tag>/
set -u
tag=${1:-prof}
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out/$tag
mkdir -p "$out"
bash $repo/tools/profile_bench.sh $tag > "$out/per_call_summary.txt" 2>&1
bash $repo/tools/profile_ns_pair.sh $tag > "$out/ns_pair.log" 2>&1
bash $repo/tools/profile_pmc.sh $tag > "$out/pmc.log" 2>&1
bash $repo/tools/profile_sq.sh $tag > "$out/sq_issue_summary.txt" 2>&1
cd $repo
python tools/kernel_bench.py --shape 3d > "$out/kernel_bench_3d.log" 2>/dev/null
python tools/kernel_bench.py --shape 2d > "$out/kernel_bench_2d.log" 2>/dev/null
python tools/kernel_bench.py --shape 3d5 > "$out/kernel_bench_3d5.log" 2>/dev/null      # cfg-5's volume: rows of 80 voxels
python tools/traffic_from_pmc.py "$out" "$out/traffic.json" > "$out/traffic_summary.txt" 2>&1
ls -la "$out"
