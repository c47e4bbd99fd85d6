#!/usr/bin/env python
"""A/B of the affine warp kernels on the GPU: run once per setting of ADVCHAIN_NO_AFFINE_BOX.

    python tools/ab/affine_ab.py [--rot DEG] [--save out.pt | --check out.pt]
Times fwd C=1/C=4 and fwd+bwd C=4 at 4x.x128x128x64 and 32x.x256x256 and (optionally) stores / compares the results.
"""
import argparse
import math
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


def make_theta(N, d, rot, dev, jitter=0.05):
    g = torch.Generator().manual_seed(5)
    theta = torch.eye(d, d + 1).repeat(N, 1, 1) + jitter * torch.randn(N, d, d + 1, generator=g)
    if rot:
        cr, sr = math.cos(math.radians(rot)), math.sin(math.radians(rot))
        R = torch.eye(d)
        R[0, 0], R[0, 1], R[1, 0], R[1, 1] = cr, -sr, sr, cr
        theta = torch.cat([R @ theta[:, :, :d], theta[:, :, d:]], dim=2)
    return theta.contiguous().to(dev)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rot", type=float, default=0.0)
    ap.add_argument("--save", default=None)
    ap.add_argument("--check", default=None)
    args = ap.parse_args()
    from advchain_amd import ops
    dev = torch.device("cuda")
    res = {}
    for tag, N, dims in (("3d", 4, (128, 128, 64)), ("2d", 32, (256, 256))):
        d = len(dims)
        g = torch.Generator().manual_seed(1)
        x1 = torch.rand(N, 1, *dims, generator=g).to(dev)
        x4 = torch.rand(N, 4, *dims, generator=g).to(dev)
        g4 = torch.rand(N, 4, *dims, generator=g).to(dev)
        theta = make_theta(N, d, args.rot, dev)
        NV = x1.numel()
        t1 = timeit(lambda: ops.affine_warp(x1, theta))
        t4 = timeit(lambda: ops.affine_warp(x4, theta))
        xa, ta = x4.clone().requires_grad_(True), theta.clone().requires_grad_(True)

        def fb():
            o = ops.affine_warp(xa, ta)
            return torch.autograd.grad(o, (xa, ta), g4)
        tfb = timeit(fb)
        tb = torch.zeros(())

        def th_only():
            o = ops.affine_warp(x4, ta)
            return torch.autograd.grad(o, (ta,), g4)
        tth = timeit(th_only)
        print("%s rot=%g box=%s  fwd C=1 %.1f us (%.2f TB/s)  fwd C=4 %.1f us (%.2f TB/s)  fwd+bwd C=4 %.1f us (%.2f TB/s)  fwd+gtheta C=4 %.1f us"
              % (tag, args.rot, os.environ.get("ADVCHAIN_NO_AFFINE_BOX") is None, t1, 8 * NV / t1 / 1e6, t4, 32 * NV / t4 / 1e6,
                 tfb, 80 * NV / tfb / 1e6, tth))
        gi, gt = fb()
        res[tag] = dict(o1=ops.affine_warp(x1, theta).cpu(), o4=ops.affine_warp(x4, theta).cpu(), gin=gi.cpu(), gth=gt.cpu())
    if args.save:
        torch.save(res, args.save)
    if args.check:
        ref = torch.load(args.check)
        for tag in res:
            for k in res[tag]:
                a, b = res[tag][k], ref[tag][k]
                print("  %s %-4s max|diff| %.3e (scale %.3e) equal=%s" % (tag, k, float((a - b).abs().max()), float(b.abs().max()), torch.equal(a, b)))


if __name__ == "__main__":
    main()
