for w in cfg4 cfg5 cfg3; do for v in "" "ADVCHAIN_SCATTER_MARCH_HMAX=4"; do
echo "== $w $v"; env $v python bench.py --workload $w --only-workload --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d.get('value'), d.get('ms_per_step'))"
done; done
