#!/bin/bash
# round 6, GPU session 14: which bounds the cfg-5 replays leave (the replay leg's violated_bounds), at two margins
set -u
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out/r06lad
mkdir -p "$out"
cd $repo
for m in 1.3 1.6; do
ADVCHAIN_PLAN_MARGIN=$m timeout 900 python bench.py --workload cfg5 --no-secondary --no-cpu-baseline > "$out/bench_cfg5_m$m.json" 2> "$out/bench_cfg5_m$m.err"
python - "$out/bench_cfg5_m$m.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print({k: d.get(k) for k in ("ms_per_step", "replayed_ms_per_step")})
r = d.get("replayed") or {}
print({k: r.get(k) for k in ("steps", "replays", "violations", "captures")})
for v in r.get("violated_bounds", []):
    print("   ", v)
PY
done
