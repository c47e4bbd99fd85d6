#!/usr/bin/env python
"""The benchmark's segmentation net (Conv(1,4,3,1,1), adv_compose_solver.py:593) under MIOpen's default (immediate) kernel
choice against torch.backends.cudnn.benchmark = True (MIOpen find) and against a channels_last model (the
batched_transpose rows around the NHWC igemm): forward, and the input-gradient backward of an ascent step."""
import sys, time
import torch

def run(sd, shape, bench, channels_last=False):
    torch.backends.cudnn.benchmark = bench
    torch.manual_seed(0)
    conv = (torch.nn.Conv2d if sd == 2 else torch.nn.Conv3d)(1, 4, 3, 1, 1).cuda().eval()
    fmt = (torch.channels_last if sd == 2 else torch.channels_last_3d) if channels_last else torch.contiguous_format
    conv = conv.to(memory_format=fmt)
    for p in conv.parameters():
        p.requires_grad_(False)
    x = torch.rand(*shape, device="cuda").requires_grad_(True)
    g = torch.rand(shape[0], 4, *shape[2:], device="cuda")
    # (the path's kernels read NCHW: what a channels_last model hands over is made contiguous, as ops._dev does)
    def fwd():
        with torch.no_grad():
            return conv(x.contiguous(memory_format=fmt)).contiguous()
    def fb():
        y = conv(x.contiguous(memory_format=fmt)).contiguous()
        y.backward(g)
        x.grad = None
    out = []
    for f in (fwd, fb):
        for _ in range(5): f()
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): f()
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / 50 * 1e3)
    print("%dD %-22s channels_last=%-5s benchmark=%-5s fwd %7.1f us   fwd+bwd(input) %7.1f us" % (sd, shape, channels_last, bench, out[0], out[1]), flush=True)

for sd, shape in ((2, (32, 1, 256, 256)), (3, (4, 1, 128, 128, 64)), (3, (8, 1, 160, 160, 80))):
    for b in (False, True):
        run(sd, shape, b)
    run(sd, shape, False, channels_last=True)
