python -m pytest tests/test_ops_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x 2>&1 | tail -3 > gpurun_out/r02_t1.log
rm -f gpurun_out/r02_kb1.log
python tools/kernel_bench.py --shape 3d --only "e" 2>/dev/null | grep -E "grid_sample fwd|compose_self fwd|halo=-1|exact" >> gpurun_out/r02_kb1.log
echo "== bwd ZC 16" >> gpurun_out/r02_kb1.log
ADVCHAIN_MARCH_ZC=16 python tools/kernel_bench.py --shape 3d --only "bwd" 2>/dev/null | grep -E "halo=-1|exact" >> gpurun_out/r02_kb1.log
cat gpurun_out/r02_t1.log gpurun_out/r02_kb1.log
