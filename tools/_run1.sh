python -m pytest tests/test_ops_gpu.py tests/test_fullsize_gpu.py tests/test_solver_gpu.py -m gpu -q -x 2>&1 | tail -6 > gpurun_out/r02_t1.log
python bench.py --workload cfg5 --steps 3 --warmup 1 --only-workload 2>/dev/null | tail -1 | cut -c1-400 > gpurun_out/r02_kb1.log
ADVCHAIN_NO_MARCH_FWD=1 ADVCHAIN_NO_MARCH_ADJOINT=1 python bench.py --workload cfg5 --steps 3 --warmup 1 --only-workload 2>/dev/null | tail -1 | cut -c1-400 >> gpurun_out/r02_kb1.log
cat gpurun_out/r02_t1.log gpurun_out/r02_kb1.log
