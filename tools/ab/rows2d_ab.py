#!/usr/bin/env python
"""Timing and bit-compare harness of the whole-row squaring backward (k_scatter_rows2d) across two builds of the library:
--save with one, --check with the other (tools/ab/build_variant.sh).  Round 5 used it for two variants: four pixels a lane
with 16-byte loads (179 VGPRs, one workgroup a CU: 65 against 52 us at 64 x 2 x 256 x 256, H = 4 -- dropped) and the own rows'
coordinate path evaluated where the row is visited for its deposits (45 us, kept); both bit-identical.

    python tools/ab/rows2d_ab.py --save /tmp/a.pt ; <other build> python tools/ab/rows2d_ab.py --check /tmp/a.pt
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def field(N, dims, px, seed, dev):
    """id + a smooth displacement of at most `px` pixels (normalised units), like a squaring's input."""
    from oracle import advchain_oracle as O
    g = torch.Generator().manual_seed(seed)
    H, W = dims
    coarse = torch.rand(N, 2, 9, 9, generator=g) * 2 - 1
    disp = torch.nn.functional.interpolate(coarse, size=dims, mode="bicubic", align_corners=True)
    disp = disp / disp.abs().amax(dim=(1, 2, 3), keepdim=True)
    scale = torch.tensor([2.0 / (W - 1), 2.0 / (H - 1)]).view(1, 2, 1, 1) * px * 0.98
    return (O.identity_grid(N, dims) + disp * scale).contiguous().to(dev)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--save", default=None)
    ap.add_argument("--check", default=None)
    args = ap.parse_args()
    from advchain_amd import ops
    dev = torch.device("cuda")
    res = {}
    for N, dims in ((64, (256, 256)), (8, (192, 192)), (5, (100, 128)), (3, (64, 512)), (2, (33, 16))):
        for px, halo in ((1.7, -2), (3.5, -4), (7.0, -8), (14.0, -16), (28.0, -32)):
            if px > min(dims) / 2:
                continue
            phi = field(N, dims, px, 11, dev)
            g = torch.Generator().manual_seed(3)
            go = torch.randn(N, 2, *dims, generator=g).to(dev)
            ws = ops._scatter_workspace(N, dims, dev)
            out = ops.raw_compose_self_bwd(go, phi, ws, chain=False, halo=halo)
            res[(N, dims, halo)] = out.cpu()
            t = timeit(lambda: ops.raw_compose_self_bwd(go, phi, ws, chain=False, halo=halo))
            print("N=%-3d %-10s halo %-4d %7.1f us" % (N, "x".join(map(str, dims)), halo, t), flush=True)
    if args.save:
        torch.save(res, args.save)
    if args.check:
        ref = torch.load(args.check)
        bad = [k for k in res if not torch.equal(res[k], ref[k])]
        print("bit-identical to %s: %s" % (args.check, "yes" if not bad else "NO: %s" % bad))
        for k in bad:
            print("   ", k, float((res[k] - ref[k]).abs().max()))


if __name__ == "__main__":
    main()
