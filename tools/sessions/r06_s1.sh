#!/bin/bash
# round 6, GPU session 1: the whole GPU suite, the default bench line, the north-star pair under rocprofv3 (row maxima folded
# into the wide scatter vs the pre-pass).   gpurun -- bash tools/sessions/r06_s1.sh
set -u
tag=r06a
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out/$tag
mkdir -p "$out"
cd $repo
export ADVCHAIN_PARITY_LOG=$out/parity_levels.txt
rm -f $ADVCHAIN_PARITY_LOG
timeout 1500 python -m pytest tests -m gpu -x -q > "$out/pytest_gpu.log" 2>&1
tail -15 "$out/pytest_gpu.log"
timeout 900 python bench.py > "$out/bench_default.json" 2> "$out/bench_default.err"
tail -c 3000 "$out/bench_default.json"; tail -5 "$out/bench_default.err"
bash tools/profile_ns_pair.sh $tag > "$out/ns_pair.log" 2>&1
ADVCHAIN_WIDE_ROWMAX_PASS=1 bash tools/profile_ns_pair.sh ${tag}_rowmaxpass > "$out/ns_pair_rowmaxpass.log" 2>&1
cat $out/ns_pair_summary.txt $repo/gpurun_out/${tag}_rowmaxpass/ns_pair_summary.txt
