python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "grid_sample or demons or morph" 2>&1 | tail -3 > gpurun_out/r02_t1.log
rm -f gpurun_out/r02_kb1.log
for zc in 4 8 16; do
  echo "== FWD ZC=$zc" >> gpurun_out/r02_kb1.log
  ADVCHAIN_FWD_MARCH_ZC=$zc python tools/kernel_bench.py --shape 3d --only "fwd" 2>/dev/null | grep -E "grid_sample fwd|compose_self fwd" >> gpurun_out/r02_kb1.log
done
cat gpurun_out/r02_t1.log gpurun_out/r02_kb1.log
