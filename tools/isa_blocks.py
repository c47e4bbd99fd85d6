#!/usr/bin/env python
"""Per basic block of a kernel in a gfx950 .s file (hipcc -S): VALU / SALU / LDS / vector-memory instruction counts, waits on
LDS (lgkmcnt) and on memory (vmcnt), barriers and branches -- where a hot loop spends its instructions and how often it stops
for an LDS round trip.   tools/isa_blocks.py file.s name-substring [min-instructions]"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
want = sys.argv[2]
floor = int(sys.argv[3]) if len(sys.argv) > 3 else 12
for m in re.finditer(r"\n(_Z\w+):\s*; @", s):
    name = m.group(1)
    if want not in name:
        continue
    body = s[m.end():]
    body = body[:body.find(".Lfunc_end")]
    print(name[:150])
    cur = ["entry", collections.Counter(), []]
    blocks = []
    for line in body.split("\n"):
        lm = re.match(r"^(\.LBB\d+_\d+):", line)
        if lm:
            blocks.append(cur)
            cur = [lm.group(1), collections.Counter(), []]
            continue
        mm = re.match(r"\s+([a-z_0-9]+)\s*(.*)", line)
        if not mm:
            continue
        op, rest = mm.group(1), mm.group(2)
        c = cur[1]
        if op.startswith("v_"):
            c["valu"] += 1
        elif op.startswith("ds_"):
            c["lds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_")):
            c["vmem"] += 1
        elif op == "s_waitcnt":
            if "lgkmcnt" in rest:
                c["wait_lds"] += 1
            if "vmcnt" in rest:
                c["wait_mem"] += 1
        elif op == "s_barrier":
            cur[2].append("BARRIER")
        elif op.startswith("s_cbranch") or op == "s_branch":
            cur[2].append(op.replace("s_cbranch_", "") + ">" + rest.split()[0].replace(".LBB", ""))
            c["salu"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
    blocks.append(cur)
    for b in blocks:
        n = sum(b[1][k] for k in ("valu", "salu", "lds", "vmem"))
        if n >= floor or "BARRIER" in b[2]:
            print("  %-12s valu %4d salu %4d lds %3d vmem %3d  waits lds %2d mem %2d  %s" % (
                b[0].replace(".LBB", ""), b[1]["valu"], b[1]["salu"], b[1]["lds"], b[1]["vmem"], b[1]["wait_lds"], b[1]["wait_mem"],
                " ".join(b[2])))
