#!/bin/bash
# The ordered kernel sequence of the LAST adversarial_training call of a short launch-by-launch bench run (rocprofv3 kernel trace).
#   tools/ab/step_sequence.sh <workload> <out.txt>
set -u
wl=${1:-cfg2}; out=${2:-gpurun_out/step_sequence_$wl.txt}
repo=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp; cd /tmp
python $repo/bench.py --workload $wl --steps ${SEQ_STEPS-2} --warmup 1 --only-workload ${SEQ_GRAPH_FLAG---no-graph} > /dev/null 2>&1
rm -rf /tmp/rp_seq
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/rp_seq -o seq -- python $repo/bench.py --workload $wl --steps ${SEQ_STEPS-2} --warmup 1 --only-workload ${SEQ_GRAPH_FLAG---no-graph} > /tmp/rp_seq.log 2>&1
f=$(find /tmp/rp_seq -name "seq_kernel_trace.csv" | head -1)
python - "$f" "$repo/$out" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# calls are separated by the random-init kernels (distribution_elementwise) -- take everything after the last big gap instead:
names = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
# last call: from the last occurrence of the normal-distribution kernel (noise init)
idx = max(i for i, n in enumerate(names) if "normal" in n[0] or "distribution" in n[0])
# walk back to the first distribution kernel of that cluster
while idx > 0 and any("distribution" in names[j][0] for j in range(max(0, idx - 12), idx)):
    idx = max(j for j in range(max(0, idx - 12), idx) if "distribution" in names[j][0])
seq = names[idx:]
t0 = seq[0][1]
with open(sys.argv[2], "w") as f:
    for n, a, b in seq:
        short = re.sub(r"^void ", "", n); short = re.sub(r"advchain::", "", short); short = re.sub(r"\((?!anonymous).*", "", short)[:100]
        f.write("%9.1f %7.1f  %s\n" % ((a - t0) / 1e3, (b - a) / 1e3, short))
print(len(seq), "kernels in the last call;", (seq[-1][2] - t0) / 1e6, "ms")
PY
