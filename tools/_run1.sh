rm -f gpurun_out/r02_kb1.log
for v in "ADVCHAIN_MARCH_C1_NW=8 ADVCHAIN_MARCH_ZC=8" "ADVCHAIN_MARCH_C1_NW=8 ADVCHAIN_MARCH_ZC=16" "ADVCHAIN_MARCH_C1_NW=4"; do
  echo "== $v" >> gpurun_out/r02_kb1.log
  env $v python tools/kernel_bench.py --shape 3d --only "grid_sample bwd C=1 (gin+ggrid) halo=-1" 2>/dev/null | grep -E "halo=-1" >> gpurun_out/r02_kb1.log
done
python - >> gpurun_out/r02_kb1.log 2>&1 <<'PY'
import sys, json; sys.path.insert(0,'.')
import torch, bench
r = bench.grid_sample3d_roofline(torch.device('cuda'))
print(json.dumps(r['levels'], indent=1))
PY
cat gpurun_out/r02_kb1.log
