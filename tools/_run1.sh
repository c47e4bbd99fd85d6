python -m pytest tests/test_ops_gpu.py tests/test_solver_gpu.py -m gpu -q -x -k "axpy or solver or golden" 2>&1 | tail -2
for w in cfg1 cfg2; do
echo "== $w"; python bench.py --workload $w --only-workload --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d.get('value'), d.get('ms_per_step'))"
done
