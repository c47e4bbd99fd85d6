python -m pytest tests/test_ops_gpu.py -m gpu -q -x 2>&1 | tail -2
for w in cfg4 cfg5; do
echo "== $w"; python bench.py --workload $w --only-workload --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d.get('value'), d.get('ms_per_step'))"
done
