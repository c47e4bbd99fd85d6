#!/bin/bash
# round 6, GPU session 3: full GPU suite with the one-launch updates and the new fixtures; cfg-2 bench A/B of the update kernel
set -u
tag=r06c
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out/$tag
mkdir -p "$out"
cd $repo
export ADVCHAIN_PARITY_LOG=$out/parity_levels.txt
rm -f $ADVCHAIN_PARITY_LOG
timeout 1500 python -m pytest tests -m gpu -q > "$out/pytest_gpu.log" 2>&1
tail -8 "$out/pytest_gpu.log"
for rep in 1 2; do
timeout 600 python bench.py --no-secondary --no-cpu-baseline --no-replay-leg > "$out/bench_cfg2_$rep.json" 2> "$out/bench_cfg2.err"
ADVCHAIN_FUSED_UPDATE_OFF=1 timeout 600 python bench.py --no-secondary --no-cpu-baseline --no-replay-leg > "$out/bench_cfg2_unfused_update_$rep.json" 2>/dev/null
python - "$out/bench_cfg2_$rep.json" "$out/bench_cfg2_unfused_update_$rep.json" <<'PY'
import json, sys
for p in sys.argv[1:]:
    d = json.load(open(p))
    print(p.split("/")[-1], {k: d.get(k) for k in ("value", "ms_per_step", "deterministic_ms_per_step", "gpu_busy_ms_per_step", "launches_per_step")})
PY
done
