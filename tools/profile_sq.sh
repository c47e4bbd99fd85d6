#!/bin/bash
# Issue / stall counters (SQ block, one pass: 8 slots) of the per-kernel micro-benchmark: which pipe a kernel spends its
# wave-cycles on.   tools/profile_sq.sh <tag>  ->  gpurun_out/<tag>/pmc_sq_kb3d.csv
set -u
tag=${1:-sq}
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
for shape in 3d 2d; do
  rm -rf /tmp/pmc_sq_$shape
  timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM \
      --kernel-trace --output-format csv -d /tmp/pmc_sq_$shape -o kb -- python $repo/tools/kernel_bench.py --shape $shape --reps 3 > /tmp/pmc_sq_$shape.log 2>&1
  python $repo/tools/pmc_summary.py /tmp/pmc_sq_$shape "$out/pmc_sq_kb$shape.csv"
done
python - "$out" <<'PY'
import csv, sys, collections
root = sys.argv[1]
for shape in ("3d", "2d"):
    rows = collections.defaultdict(dict)
    for r in csv.DictReader(open("%s/pmc_sq_kb%s.csv" % (root, shape))):
        rows[r["kernel"]][r["counter"]] = float(r["mean_value"])
    print("== %s: share of wave-cycles (quad-cycle units) per kernel" % shape)
    print("%-52s %9s %7s %7s %7s %7s %7s %9s" % ("kernel", "wavecyc", "waitI%", "valu%", "vmem%", "lds%", "salu%", "vmem_inst"))
    for k, c in sorted(rows.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
        w = c.get("SQ_WAVE_CYCLES", 0)
        if w <= 0 or not (k.startswith("k_") or "advchain" in k):
            continue
        pct = lambda n: 100.0 * c.get(n, 0) / w
        print("%-52s %9.0f %7.1f %7.1f %7.1f %7.1f %7.1f %9.0f" % (k[:52], w, pct("SQ_WAIT_INST_ANY"), pct("SQ_ACTIVE_INST_VALU"),
              pct("SQ_ACTIVE_INST_VMEM"), pct("SQ_ACTIVE_INST_LDS"), pct("SQ_ACTIVE_INST_SCA"), c.get("SQ_INSTS_VMEM", 0)))
PY
