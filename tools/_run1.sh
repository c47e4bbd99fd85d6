for v in "" "ADVCHAIN_GTILE_T0=4" "ADVCHAIN_GTILE_T1=16" "ADVCHAIN_GTILE_H3=2" "ADVCHAIN_GTILE_T0=4 ADVCHAIN_GTILE_H3=2" "ADVCHAIN_GTILE_T0=1 ADVCHAIN_GTILE_T1=16"; do
echo "== cfg5 $v"; env $v python bench.py --workload cfg5 --only-workload --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d.get('value'), d.get('ms_per_step'))"
done
