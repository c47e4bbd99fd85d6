#!/bin/bash
# round 6, GPU session 2: the one-launch parameter updates (tests + bench), the occupancy experiment behind the f1-in-3D decision
# (the two sub-voxel chain kernels with the workgroups-per-CU a fused two-level kernel's LDS footprint would leave them)
set -u
tag=r06b
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out/$tag
mkdir -p "$out/f1_3d"
cd $repo
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_solver_gpu.py tests/test_graph_gpu.py tests/test_dist_gpu.py tests/test_deterministic_gpu.py -m gpu -x -q > "$out/pytest_subset.log" 2>&1
tail -6 "$out/pytest_subset.log"
timeout 600 python bench.py --no-secondary --no-cpu-baseline > "$out/bench_cfg2.json" 2> "$out/bench_cfg2.err"
python - "$out/bench_cfg2.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print({k: d.get(k) for k in ("value", "ms_per_step", "replayed_ms_per_step", "deterministic_ms_per_step", "gpu_busy_ms_per_step", "launches_per_step")})
PY
ADVCHAIN_FUSED_UPDATE_OFF=1 timeout 600 python bench.py --no-secondary --no-cpu-baseline --no-replay-leg > "$out/bench_cfg2_unfused_update.json" 2>/dev/null
python - "$out/bench_cfg2_unfused_update.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("per-transform updates:", {k: d.get(k) for k in ("value", "ms_per_step", "gpu_busy_ms_per_step", "launches_per_step")})
PY
for pad in 0 24 40 100; do
  for b in 4 8; do
    ADVCHAIN_MARCH_LDS_PAD=$pad python tools/kernel_bench.py --shape 3d --batch $b --reps 30 --only "compose_self" > "$out/f1_3d/kb3d_n${b}_ldspad${pad}.log" 2>&1
    echo "== LDS pad $pad KiB, N = $b"; grep -E "compose_self fwd|exact bound" "$out/f1_3d/kb3d_n${b}_ldspad${pad}.log"
  done
done
