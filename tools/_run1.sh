rm -f gpurun_out/r02_kb1.log
for v in "ADVCHAIN_MARCH_SELF_RPW=8 ADVCHAIN_MARCH_ZC=8" "ADVCHAIN_MARCH_SELF_RPW=8 ADVCHAIN_MARCH_ZC=16" "ADVCHAIN_MARCH_SELF_RPW=2 ADVCHAIN_MARCH_ZC=16" "ADVCHAIN_MARCH_SELF_RPW=1 ADVCHAIN_MARCH_ZC=16"; do
  echo "== $v" >> gpurun_out/r02_kb1.log
  env $v python tools/kernel_bench.py --shape 3d --only "compose_self bwd halo=-1" 2>/dev/null | grep -E "exact" >> gpurun_out/r02_kb1.log
done
python tools/kernel_bench.py --shape 3d --only "fwd" 2>/dev/null | grep -E "grid_sample fwd|compose_self fwd" >> gpurun_out/r02_kb1.log
cat gpurun_out/r02_kb1.log
