#!/usr/bin/env python
"""2D image-warp backward (C = 1 and 4, grad_in + grad_grid) at 16..32 px: the window scatter (hint 16) against the whole-row
scatter with an exact bound of 24 / 32 px (needs rows2d_tile to admit C == 1 up to 32).
Measured in round 5, 32 x 1 x 256 x 256: 32.4 against 36.6 us at 21 px, 33.6 against 41.9 at 28 px -- the window scatter stays."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tools.ab.rows2d_ab import field, timeit

def main():
    from advchain_amd import ops
    dev = torch.device("cuda")
    for N, dims in ((32, (256, 256)), (4, (192, 192))):
        for px, exact in ((14.0, -16), (21.0, -24), (28.0, -32)):
            grid = field(N, dims, px, 11, dev)
            for C in (1, 4):
                g = torch.Generator().manual_seed(3)
                go = torch.randn(N, C, *dims, generator=g).to(dev)
                inp = torch.randn(N, C, *dims, generator=g).to(dev)
                outs = {}
                for name, halo in (("window", 16), ("rows", exact)):
                    fn = lambda: ops.raw_grid_sample_bwd(go, inp, grid, 0, 0, False, True, True, halo)
                    outs[name] = fn()
                    t = timeit(fn)
                    print("N=%-3d %-8s C=%d %5.1f px %-7s halo %-4d %7.1f us" % (N, "x".join(map(str, dims)), C, px, name, halo, t), flush=True)
                a, b = outs["window"], outs["rows"]
                print("      max diff gin %.3g ggrid %.3g (scale %.3g)" % (float((a[0] - b[0]).abs().max()), float((a[1] - b[1]).abs().max()), float(a[0].abs().max())))

if __name__ == "__main__":
    main()
