#!/usr/bin/env python
"""Per-call summary of a rocprofv3 kernel_stats CSV (drops MIOpen's one-off find/tuning kernels)."""
import csv
import sys


def main(path, calls, top=24):
    rows = list(csv.DictReader(open(path)))
    items, tot, n = [], 0.0, 0
    for r in rows:
        name = r["Name"]
        avg = float(r["AverageNs"])
        if "naive_conv" in name or "kernel_batched_gemm_xdlops_bwd_weight" in name:
            continue
        if ("miopenSp3AsmConv" in name or name.startswith("void ck::")) and avg > 1e6:
            continue
        t = float(r["TotalDurationNs"])
        tot += t
        n += int(r["Calls"])
        items.append((t, name.replace("void advchain::", "").split("(")[0][:58], int(r["Calls"]), avg / 1e3))
    print("%s: GPU busy %.2f ms/call, %.0f launches/call" % (path, tot / 1e6 / calls, n / calls))
    for t, nm, c, a in sorted(items, reverse=True)[:top]:
        print("   %-58s %6.2f ms/call %6.1f launches/call  avg %8.1f us" % (nm, t / 1e6 / calls, c / calls, a))


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 24)
