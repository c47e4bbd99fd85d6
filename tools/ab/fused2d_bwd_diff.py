#!/usr/bin/env python
"""Where does the fused 2D backward differ from the per-squaring launches?  (debug aid)"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests.test_fused2d_gpu import _chain, _phi0, rand, DEV
from advchain_amd import _lib, ops
k = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dims = (40, 64)
n, N = 7, 1
phi0 = _phi0(N, dims, 0.9 / 2 ** (k - 1) * 0.95, 31 + k)
fields, pos, dm, _ = _chain(phi0, n, None, False)
dm = dm.tolist()
halos = [ops.squaring_halo(dm[m], 2) for m in range(n - 1, -1, -1)]
print("halos", halos)
gpos = rand((N, 2) + tuple(dims), 41).to(DEV)
ws = ops._scatter_workspace(N, dims, DEV)
phis = [phi0] + list(fields.unbind(0))
g = gpos
for i, phi in enumerate(reversed(phis)):
    g = ops.raw_compose_self_bwd(g, phi, ws, chain=i > 0, halo=halos[i])
out, scratch = torch.full_like(gpos, float("nan")), torch.full_like(gpos, float("nan"))
lib = _lib.load()
_lib.check(lib.advchain_expo_chain_bwd(ops._ptr(gpos), ops._ptr(phi0), ops._ptr(fields), ops._ptr(out), ops._ptr(scratch),
                                       ops._ptr(ws), (ctypes.c_int32 * n)(*halos), N, 2, _lib.dims_array(dims), n, ops._stream()), "bwd")
torch.cuda.synchronize()
d = (out - g).abs()
print("max diff %.3e of %.3e; differing %d of %d; nan %d" % (float(torch.nan_to_num(d, nan=0).max()), float(g.abs().max()), int((d > 0).sum()), d.numel(), int(torch.isnan(out).sum())))
nz = (d > 0).nonzero()
if len(nz):
    print("rows", sorted(set(nz[:, 2].tolist()))[:40])
    print("cols", sorted(set(nz[:, 3].tolist()))[:70])
    big = (d > 1e-3 * float(g.abs().max())).nonzero()
    print("big diffs", len(big), big[:10].tolist())
