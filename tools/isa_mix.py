#!/usr/bin/env python
"""Static instruction mix of the kernels in a gfx950 .s file (hipcc --save-temps):  tools/isa_mix.py file.s [name-substring]"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
want = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"\n(_Z\w+):\s*; @", s):
    name = m.group(1)
    if want not in name:
        continue
    body = s[m.end():]
    body = body[:body.find("s_endpgm")]
    cnt, ops = collections.Counter(), collections.Counter()
    for line in body.split("\n"):
        mm = re.match(r"\s+([a-z_0-9]+)\s", line)
        if not mm:
            continue
        op = mm.group(1)
        kind = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else \
            "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other"
        cnt[kind] += 1
        if kind == "valu":
            ops[op] += 1
    print(name[:110], dict(cnt))
    print("    ", ops.most_common(18))
