#!/usr/bin/env python
"""GPU idle gaps from a rocprofv3 --kernel-trace CSV: where the device waits for the host (syncs, launch overhead)."""
import csv
import sys
from collections import defaultdict


def short(n):
    return n.replace("void advchain::", "").replace("advchain::", "").split("(")[0][:48]


def main(path, skip_frac=0.5, thresh_us=15.0):
    rows = list(csv.DictReader(open(path)))
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
    ev = ev[int(len(ev) * skip_frac):]          # steady state only
    busy = sum(e - s for s, e, _ in ev)
    span = ev[-1][1] - ev[0][0]
    gaps = defaultdict(lambda: [0.0, 0])
    small = 0.0
    for (s0, e0, n0), (s1, e1, n1) in zip(ev, ev[1:]):
        g = (s1 - e0) / 1e3
        if g > thresh_us:
            k = short(n0) + "  ->  " + short(n1)
            gaps[k][0] += g
            gaps[k][1] += 1
        elif g > 0:
            small += g
    print("span %.2f ms, busy %.2f ms (%.1f%%), gaps > %.0f us: %.2f ms, smaller gaps: %.2f ms over %d kernels" % (
        span / 1e6, busy / 1e6, 100.0 * busy / span, thresh_us, sum(v[0] for v in gaps.values()) / 1e3, small / 1e3, len(ev)))
    for k, (t, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:25]:
        print("  %8.1f us  x%-4d avg %6.1f  %s" % (t, c, t / c, k))


if __name__ == "__main__":
    main(sys.argv[1])
