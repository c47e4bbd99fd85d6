#!/bin/bash
# Average kernel durations (rocprofv3 kernel trace) of a command, filtered by a name pattern:
#   tools/ab/rocprof_avg.sh <pattern> -- <command...>
pat=$1; shift 2
export TMPDIR=/tmp
d=/tmp/rp_avg_$$
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $d -o t -- "$@" > /dev/null 2>&1)
python - "$pat" "$(find $d -name 't_kernel_stats.csv' | head -1)" <<'PY'
import csv, re, sys
pat, path = re.compile(sys.argv[1]), sys.argv[2]
for r in csv.DictReader(open(path)):
    if pat.search(r["Name"]):
        print("   %-70s %6d calls  avg %8.1f us" % (r["Name"][:70], int(r["Calls"]), float(r["AverageNs"]) / 1e3))
PY
rm -rf $d
