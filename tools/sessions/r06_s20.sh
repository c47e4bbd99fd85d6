#!/bin/bash
# round 6, GPU session 20: what the driver runs at round end, on a fresh box: build check, smoke(), the GPU suite, the default bench
set -u
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out/r06w
mkdir -p "$out"
cd $repo
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout 1500 python -m pytest tests -x -q -m gpu > "$out/pytest_gpu.log" 2>&1
tail -2 "$out/pytest_gpu.log"
( time python bench.py > "$out/bench_default.json" 2> "$out/bench_default.err" ) 2>&1 | grep real
python - "$out/bench_default.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print({k: d.get(k) for k in ("metric", "value", "unit", "ms_per_step", "n_gpus", "steps", "warmup", "vs_baseline", "dtype")})
print(d["roofline"]["frac"], d["cpu_baseline"])
PY
