mkdir -p gpurun_out/r02c
python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/r02c/pytest_gpu.log
python bench.py > gpurun_out/r02c/bench_cfg2_default.json 2> gpurun_out/r02c/bench_cfg2_default.err
python bench.py --workload cfg1 --steps 50 --warmup 5 --only-workload > gpurun_out/r02c/bench_cfg1.json 2>/dev/null
bash tools/profile_bench.sh r02c > gpurun_out/r02c/per_call_summary.txt 2>&1
python tools/kernel_bench.py --shape 3d > gpurun_out/r02c/kernel_bench_3d.log 2>/dev/null
python tools/kernel_bench.py --shape 2d > gpurun_out/r02c/kernel_bench_2d.log 2>/dev/null
bash tools/profile_pmc.sh r02c > gpurun_out/r02c/pmc.log 2>&1
python tools/traffic_from_pmc.py gpurun_out/r02c gpurun_out/r02c/traffic.json > gpurun_out/r02c/traffic_summary.txt 2>&1
tail -3 gpurun_out/r02c/pytest_gpu.log; tail -3 gpurun_out/r02c/traffic_summary.txt
