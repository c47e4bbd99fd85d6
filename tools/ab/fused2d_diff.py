#!/usr/bin/env python
"""Where do the fused 2D squarings differ from the per-squaring launches?  (debug aid)"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests.test_fused2d_gpu import _chain, _phi0
n = 8
phi0 = _phi0(3, (256, 256), 0.1, 14)
ref = _chain(phi0, n, None, False)
out = _chain(phi0, n, [1] * 4 + [0] * 4, True)
print("flag", out[3])
for m in range(n - 1):
    d = (out[0][m] - ref[0][m]).abs()
    nz = (d > 0).nonzero()
    print("level", m + 1, "max diff %.3e" % float(d.max()), "differing", int((d > 0).sum()), "of", d.numel(),
          "first", nz[:4].tolist(), "x range", (int(nz[:, 3].min()), int(nz[:, 3].max())) if len(nz) else None,
          "y range", (int(nz[:, 2].min()), int(nz[:, 2].max())) if len(nz) else None)
