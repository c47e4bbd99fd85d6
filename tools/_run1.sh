python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "window or vs_aten" 2>&1 | tail -2
export TMPDIR=/tmp; cd /tmp
rm -rf /tmp/rp1; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp1 -o c2 -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --only-workload > /tmp/b.log 2>&1
python $GRAFT_REPO_ROOT/tools/stats_per_call.py $(find /tmp/rp1 -name "c2_kernel_stats.csv" | head -1) 14 60 | grep -E "window|busy"
