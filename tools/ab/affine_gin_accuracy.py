#!/usr/bin/env python
"""grad_in of the affine warp: owner-computes LDS scatter (default) and lattice gather (ADVCHAIN_NO_AFFINE_BOX_GIN=1, run
the script twice) against float64 autograd of F.affine_grid + F.grid_sample on the CPU."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from advchain_amd import ops  # noqa: E402

for dims in ((64, 96), (32, 40, 64), (128, 128, 64)):
    d = len(dims)
    g = torch.Generator().manual_seed(3)
    N, C = 2, 4
    x = torch.rand(N, C, *dims, generator=g)
    w = torch.randn(N, C, *dims, generator=g) ** 3          # heavy-tailed upstream gradient
    theta = torch.eye(d, d + 1).repeat(N, 1, 1) + 0.08 * torch.randn(N, d, d + 1, generator=g)
    xd = x.double().requires_grad_(True)
    out = F.grid_sample(xd, F.affine_grid(theta.double(), xd.size(), align_corners=True), align_corners=True)
    (out * w.double()).sum().backward()
    ref = xd.grad
    xg = x.cuda().requires_grad_(True)
    o = ops.affine_warp(xg, theta.cuda())
    (o * w.cuda()).sum().backward()
    err = (xg.grad.cpu().double() - ref).abs()
    print("%s gin=%s: max err %.3e  mean err %.3e  (|ref| max %.3e, |w| max %.2e)  fwd err %.3e" % (
        dims, "gather" if os.environ.get("ADVCHAIN_NO_AFFINE_BOX_GIN") else "scatter", float(err.max()), float(err.mean()),
        float(ref.abs().max()), float(w.abs().max()), float((o.detach().cpu().double() - out.detach()).abs().max())))
