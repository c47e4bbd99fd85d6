#!/usr/bin/env python
"""Where the time of the 3D march scatter goes at the north-star shape: grid_sample bwd (C = 1) on a given field with and
without grad_grid (A/B knob: ADVCHAIN_NO_SCATTER_MARCH_WIDE)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from advchain_amd import _lib, ops

def main():
    q = torch.load(sys.argv[1])[sys.argv[2]].cuda()
    C = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    lib = _lib.load()
    N, dims = q.shape[0], tuple(q.shape[2:])
    torch.manual_seed(0)
    x, go = torch.rand(N, C, *dims, device="cuda"), torch.rand(N, C, *dims, device="cuda")
    halo = ops.warp_halo(ops.grid_displacement(q), 3)
    gin, ggrid = torch.empty_like(x), torch.empty_like(q)
    ws = ops._scatter_workspace(N, dims, x.device)
    da = _lib.dims_array(dims)
    for gg in (True, False):
        def run():
            _lib.check(lib.advchain_grid_sample_bwd(ops._ptr(go), ops._ptr(x), ops._ptr(q), ops._ptr(gin), ops._ptr(ggrid) if gg else None,
                                                    ops._ptr(ws), N, C, 3, da, da, 0, 0, 1, halo, ops._stream()), "bwd")
        for _ in range(3): run()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        torch.cuda.synchronize(); e[0].record()
        for _ in range(30): run()
        e[1].record(); torch.cuda.synchronize()
        print("halo %d C %d grad_grid %s: %.1f us  (env %s)" % (halo, C, gg, e[0].elapsed_time(e[1]) / 30 * 1e3,
              {k: v for k, v in os.environ.items() if k.startswith("ADVCHAIN_")}))
main()
