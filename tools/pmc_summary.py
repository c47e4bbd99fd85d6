#!/usr/bin/env python
"""Summarise a rocprofv3 counter_collection CSV: mean counter value per kernel name (and call count)."""
import csv
import glob
import sys
from collections import defaultdict


def main(root, out):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"].replace("void advchain::", "").split("(")[0][:80], r["Counter_Name"])
            acc[k][0] += float(r["Counter_Value"])
            acc[k][1] += 1
    with open(out, "w") as fo:
        fo.write("kernel,counter,calls,mean_value\n")
        for (k, c), (s, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
            fo.write('"%s",%s,%d,%.1f\n' % (k, c, n, s / n))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
