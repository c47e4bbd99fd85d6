python -m pytest tests/test_ops_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x -k "gauss or demons or golden or whole_solver or one_step" 2>&1 | tail -3
for v in "" "ADVCHAIN_NO_GAUSS_ZMARCH=1 ADVCHAIN_NO_GAUSS_RIW2=1"; do for w in cfg5 cfg3; do
echo "== $w $v"; env $v python bench.py --workload $w --only-workload --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d.get('value'), d.get('ms_per_step'))"
done; done
python tools/kernel_bench.py --shape 3d --only gauss 2>/dev/null | grep gauss
