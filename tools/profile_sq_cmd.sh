#!/bin/bash
# Issue / stall counters (SQ block, own pass, 8 slots) of an arbitrary command:
#   tools/profile_sq_cmd.sh <tag> <kernel-name filter> <command ...>  ->  gpurun_out/<tag>/pmc_sq.csv + a printed table
# WAIT_ANY = wave parked at s_waitcnt / barrier, WAIT_INST_ANY = issue stall, ACTIVE_INST_* = issuing to that pipe
# (quad-cycle units, disjoint: WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES, MI355X_MICROARCH.md).
set -u
tag=$1; filt=$2; shift 2
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/pmc_sq_cmd
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAVES \
    --kernel-trace --output-format csv -d /tmp/pmc_sq_cmd -o p -- "$@" > "$out/cmd.log" 2>&1
python $repo/tools/pmc_summary.py /tmp/pmc_sq_cmd "$out/pmc_sq.csv"
python - "$out/pmc_sq.csv" "$filt" <<'PY'
import csv, sys, collections
rows = collections.defaultdict(dict)
calls = {}
for r in csv.DictReader(open(sys.argv[1])):
    rows[r["kernel"]][r["counter"]] = float(r["mean_value"])
    calls[r["kernel"]] = int(r["calls"])
print("%-56s %5s %10s %6s %6s %6s %6s %9s %8s" % ("kernel", "calls", "wavecyc", "park%", "stall%", "valu%", "lds%", "valu/wave", "waves"))
for k, c in sorted(rows.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    w = c.get("SQ_WAVE_CYCLES", 0)
    if w <= 0 or sys.argv[2] not in k:
        continue
    pct = lambda n: 100.0 * c.get(n, 0) / w
    print("%-56s %5d %10.0f %6.1f %6.1f %6.1f %6.1f %9.0f %8.0f" % (k[:56], calls[k], w, pct("SQ_WAIT_ANY"), pct("SQ_WAIT_INST_ANY"),
          pct("SQ_ACTIVE_INST_VALU"), pct("SQ_ACTIVE_INST_LDS"), c.get("SQ_INSTS_VALU", 0) / max(1.0, c.get("SQ_WAVES", 1)), c.get("SQ_WAVES", 0)))
PY
