#!/bin/bash
# round 6, GPU session 4: the joined tap block of the 3D forward march (A/B of three library builds), the fused smoothing +
# upsampling launch, the wider slot reduction -- parity first, then timings
set -u
tag=r06d
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out/$tag
mkdir -p "$out"
cd $repo
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_fullsize_gpu.py tests/test_fused2d_gpu.py tests/test_solver_gpu.py tests/test_ride_gpu.py -m gpu -q > "$out/pytest_subset.log" 2>&1
tail -5 "$out/pytest_subset.log"
lib=advchain_amd/csrc/libadvchain_hip.so
cp $lib /tmp/lib_keep.so
for v in old join2 join3 old join2; do
  cp scratch/ab/lib_fwd_$v.so $lib
  echo "== forward march build: $v"
  for b in 4 8; do
    python tools/kernel_bench.py --shape 3d --batch $b --reps 40 --only " fwd" 2>/dev/null | grep -E "grid_sample fwd|compose_self fwd" | sed "s/^/N=$b  /"
  done
  python tools/kernel_bench.py --shape 3d5 --reps 30 --only " fwd" 2>/dev/null | grep -E "grid_sample fwd|compose_self fwd" | sed "s/^/3d5  /"
done > "$out/fwd_join_ab.txt" 2>&1
cat "$out/fwd_join_ab.txt"
for v in old join2; do
  cp scratch/ab/lib_fwd_$v.so $lib
  for w in cfg3 cfg5; do
    python bench.py --workload $w --only-workload --steps 6 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v $w', d['ms_per_step'], d['value'])"
  done
done | tee "$out/fwd_join_e2e.txt"
cp /tmp/lib_keep.so $lib
timeout 600 python bench.py --no-secondary --no-cpu-baseline --no-replay-leg > "$out/bench_cfg2.json" 2>/dev/null
python - "$out/bench_cfg2.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print({k: d.get(k) for k in ("value", "ms_per_step", "deterministic_ms_per_step", "gpu_busy_ms_per_step", "launches_per_step")})
PY
