python -m pytest tests/test_ops_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x 2>&1 | tail -8 > gpurun_out/r02_t1.log
rm -f gpurun_out/r02_kb1.log
for dbg in 0 16 32 48 128 240; do
  echo "== ZC=8 dbg=$dbg" >> gpurun_out/r02_kb1.log
  ADVCHAIN_MARCH_DEBUG=$dbg ADVCHAIN_MARCH_ZC=8 ADVCHAIN_MARCH_SELF_RPW=2 python tools/kernel_bench.py --shape 3d --only "bwd" 2>/dev/null | grep -E "halo=-1|exact" >> gpurun_out/r02_kb1.log
done
echo "== ZC=8 self rpw1 c1 rpw1" >> gpurun_out/r02_kb1.log
ADVCHAIN_MARCH_ZC=8 ADVCHAIN_MARCH_SELF_RPW=1 ADVCHAIN_MARCH_C1_RPW=1 python tools/kernel_bench.py --shape 3d --only "bwd" 2>/dev/null | grep -E "halo=-1|exact" >> gpurun_out/r02_kb1.log
echo "== ZC=16 " >> gpurun_out/r02_kb1.log
ADVCHAIN_MARCH_ZC=16 ADVCHAIN_MARCH_SELF_RPW=2 python tools/kernel_bench.py --shape 3d --only "bwd" 2>/dev/null | grep -E "halo=-1|exact" >> gpurun_out/r02_kb1.log
cat gpurun_out/r02_t1.log gpurun_out/r02_kb1.log
