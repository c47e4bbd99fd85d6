python -m pytest tests/test_ops_gpu.py tests/test_solver_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x 2>&1 | tail -4 > gpurun_out/r02_t1.log
rm -f gpurun_out/r02_kb1.log
for v in "" "ADVCHAIN_STREAM2D_YC=8" "ADVCHAIN_STREAM2D_YC=32" "ADVCHAIN_NO_STREAM2D=1"; do
  echo "== $v" >> gpurun_out/r02_kb1.log
  env $v python tools/kernel_bench.py --shape 2d --only "compose_self bwd" 2>/dev/null | grep -E "compose_self" >> gpurun_out/r02_kb1.log
done
python bench.py --only-workload 2>/dev/null | tail -1 | cut -c1-330 >> gpurun_out/r02_kb1.log
cat gpurun_out/r02_t1.log gpurun_out/r02_kb1.log
