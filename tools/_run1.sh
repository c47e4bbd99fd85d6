python -m pytest tests/test_ops_gpu.py tests/test_fullsize_gpu.py -m gpu -q -k "gauss or affine or demons or morph or linear_kernels or one_ascent" 2>&1 | tail -6 > gpurun_out/r02_t1.log
rm -f gpurun_out/r02_kb1.log
for v in "" "ADVCHAIN_NO_GAUSS_FUSED=1" "ADVCHAIN_GAUSS_ZC=32" "ADVCHAIN_GAUSS_ZC=8"; do
  echo "== $v" >> gpurun_out/r02_kb1.log
  env $v python tools/kernel_bench.py --shape 3d --only "gauss" 2>/dev/null | grep -E "gauss" >> gpurun_out/r02_kb1.log
  env $v python tools/kernel_bench.py --shape 2d --only "gauss" 2>/dev/null | grep -E "gauss" >> gpurun_out/r02_kb1.log
done
cat gpurun_out/r02_t1.log gpurun_out/r02_kb1.log
