#!/usr/bin/env python
"""Per kernel: vector-memory loads vs `s_waitcnt vmcnt(0)` in the gfx950 ISA (hipcc --save-temps).  A kernel whose
loads are nearly all followed by a full wait makes one serial memory round trip per load -- usually a conditional
load (`if (ok) acc += p[i] * w`) that the compiler wrapped in its own branch.

    tools/isa_wait_audit.py [file.hip ...]      (default: every advchain_amd/csrc/*.hip)
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def audit(src):
    with tempfile.TemporaryDirectory() as tmp:
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
               "-I" + os.path.join(ROOT, "advchain_amd", "csrc"), "-c", src, "-o", os.path.join(tmp, "x.o"), "--save-temps"]
        subprocess.run(cmd, cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        asm = glob.glob(os.path.join(tmp, "*gfx950.s"))[0]
        name, rows, cur = None, [], None
        for line in open(asm):
            m = re.match(r"^(_Z\w+):", line)
            if m:
                name = m.group(1)
                cur = {"name": name, "loads": 0, "full_waits": 0, "branches": 0}
                rows.append(cur)
                continue
            if cur is None:
                continue
            if line.startswith(".Lfunc_end"):      # (a kernel may contain several s_endpgm: early exits)
                cur = None
                continue
            if re.search(r"\b(global|buffer|flat)_load", line):
                cur["loads"] += 1
            elif "s_waitcnt vmcnt(0)" in line:
                cur["full_waits"] += 1
            elif "s_cbranch" in line:
                cur["branches"] += 1
        return rows


def demangle(n):
    try:
        return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", n], capture_output=True, text=True).stdout.strip().split("(")[0]
    except OSError:
        return n


def main():
    files = [os.path.abspath(f) for f in sys.argv[1:]] or sorted(glob.glob(os.path.join(ROOT, "advchain_amd", "csrc", "*.hip")))
    for f in files:
        for r in audit(f):
            if r["loads"] >= 4 and r["full_waits"] * 3 >= r["loads"]:
                print("%-22s loads %3d  full waits %3d  branches %3d  %s" % (os.path.basename(f), r["loads"], r["full_waits"],
                                                                          r["branches"], demangle(r["name"])[:110]))


if __name__ == "__main__":
    main()
