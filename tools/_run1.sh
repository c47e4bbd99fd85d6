export TMPDIR=/tmp; cd /tmp
for w in cfg5; do
rm -rf /tmp/rp1; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp1 -o c2 -- python $GRAFT_REPO_ROOT/bench.py --workload $w --steps 3 --warmup 1 --only-workload > /tmp/b_$w.log 2>&1
python $GRAFT_REPO_ROOT/tools/stats_per_call.py $(find /tmp/rp1 -name "c2_kernel_stats.csv" | head -1) 4 30
done
