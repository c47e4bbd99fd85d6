#!/bin/bash
# round 6, GPU session 13: the anatomy ladder behind a hipGraph replay -- the graph tests, then cfg-5 with its replay leg
set -u
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out/r06lad
mkdir -p "$out"
cd $repo
timeout 900 python -m pytest tests/test_graph_gpu.py tests/test_dist_gpu.py -m gpu -q -x > "$out/pytest_graph.log" 2>&1
tail -15 "$out/pytest_graph.log"
timeout 900 python bench.py --workload cfg5 --no-secondary --no-cpu-baseline > "$out/bench_cfg5.json" 2> "$out/bench_cfg5.err"
tail -3 "$out/bench_cfg5.err"
python - "$out/bench_cfg5.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print({k: d.get(k) for k in ("value", "ms_per_step", "replayed_ms_per_step", "replayed", "deterministic_ms_per_step", "gpu_busy_ms_per_step", "launches_per_step", "host_gap_ms_per_step")})
PY
