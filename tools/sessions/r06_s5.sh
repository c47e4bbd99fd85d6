#!/bin/bash
# round 6, GPU session 5: the whole GPU suite on the final state, then everything profiles/r06 holds (tools/profile_all.sh) and the
# default bench line
set -u
tag=r06v
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out/$tag
mkdir -p "$out"
cd $repo
export ADVCHAIN_PARITY_LOG=$out/parity_levels.txt
rm -f $ADVCHAIN_PARITY_LOG
timeout 1500 python -m pytest tests -m gpu -q > "$out/pytest_gpu.log" 2>&1
tail -6 "$out/pytest_gpu.log"
timeout 900 python bench.py > "$out/bench_default.json" 2> "$out/bench_default.err"
python - "$out/bench_default.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print({k: d.get(k) for k in ("value", "ms_per_step", "replayed_ms_per_step", "deterministic_ms_per_step", "gpu_busy_ms_per_step", "launches_per_step", "host_gap_ms_per_step")})
print(d["roofline"])
for k, v in d["other_workloads"].items():
    print(k, v["value"], v["ms_per_step"], v.get("deterministic_ms_per_step"), v.get("launches_per_step"), v["roofline"]["frac"] if v["roofline"] else None)
for k, v in d["roofline_grid_sample3d"]["levels"].items():
    print(k, v["frac"], v["trace"], v["events"])
PY
timeout 300 python bench.py --workload cfg1 --steps 50 --warmup 5 --only-workload > "$out/bench_cfg1.json" 2>/dev/null
timeout 300 python bench.py --workload cfg1 --steps 50 --warmup 5 --only-workload --graph > "$out/bench_cfg1_graph.json" 2>/dev/null
python - "$out/bench_cfg1.json" "$out/bench_cfg1_graph.json" <<'PY'
import json, sys
for p in sys.argv[1:]:
    d = json.load(open(p)); print(p.split("/")[-1], d["ms_per_step"], d["value"], d["config"]["dispatch"])
PY
bash tools/profile_all.sh $tag > "$out/profile_all.log" 2>&1
cat $out/ns_pair_summary.txt; head -3 $out/per_call_summary.txt; grep "_kernel_stats.csv" $out/per_call_summary.txt
