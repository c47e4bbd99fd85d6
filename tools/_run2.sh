python -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/r02_tfull.log
bash tools/profile_bench.sh r02b > gpurun_out/r02b_summary.txt 2>&1
bash tools/profile_pmc.sh r02b > gpurun_out/r02b_pmc.log 2>&1
python tools/traffic_from_pmc.py gpurun_out/r02b gpurun_out/r02b/traffic.json > gpurun_out/r02b/traffic_summary.txt 2>&1
cat gpurun_out/r02_tfull.log; cat gpurun_out/r02b/traffic_summary.txt
