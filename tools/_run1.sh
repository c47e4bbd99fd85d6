python -m pytest tests/test_ops_gpu.py tests/test_fullsize_gpu.py tests/test_solver_gpu.py -m gpu -q -x 2>&1 | tail -8 > gpurun_out/r02_t1.log
rm -f gpurun_out/r02_kb1.log
for zc in 8 16 32; do
  echo "== FWD ZC=$zc" >> gpurun_out/r02_kb1.log
  ADVCHAIN_FWD_MARCH_ZC=$zc python tools/kernel_bench.py --shape 3d --only "fwd" 2>/dev/null | grep -E "grid_sample fwd|compose_self fwd" >> gpurun_out/r02_kb1.log
done
echo "== old fwd" >> gpurun_out/r02_kb1.log
ADVCHAIN_NO_MARCH_FWD=1 python tools/kernel_bench.py --shape 3d --only "fwd" 2>/dev/null | grep -E "grid_sample fwd|compose_self fwd" >> gpurun_out/r02_kb1.log
cat gpurun_out/r02_t1.log gpurun_out/r02_kb1.log
