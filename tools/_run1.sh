python -m pytest tests/test_ops_gpu.py tests/test_solver_gpu.py -m gpu -q -x 2>&1 | tail -2
for w in cfg5 cfg4 cfg3; do for v in "" "ADVCHAIN_FWD_MARCH_ALWAYS=1"; do
echo "== $w $v"; env $v python bench.py --workload $w --only-workload --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d.get('value'), d.get('ms_per_step'))"
done; done
