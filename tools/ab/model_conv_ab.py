#!/usr/bin/env python
"""The benchmark's user model (Conv{2,3}d(1, 4, 3, 1, 1)) on stock MIOpen: forward and input-gradient backward
with torch.backends.cudnn.benchmark off / on (immediate mode vs. MIOpen's find).  us per call, events.

    python tools/ab/model_conv_ab.py
"""
import sys
import time

import torch


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    dev = torch.device("cuda")
    for bench in (False, True):
        torch.backends.cudnn.benchmark = bench
        for name, shape in (("cfg2 32x1x256x256", (32, 1, 256, 256)), ("cfg2 paired 64", (64, 1, 256, 256)),
                            ("cfg1 4x1x192x192", (4, 1, 192, 192)),
                            ("cfg3 4x1x128x128x64", (4, 1, 128, 128, 64)), ("cfg5 4x1x160x160x80", (4, 1, 160, 160, 80))):
            torch.manual_seed(0)
            conv = (torch.nn.Conv3d if len(shape) == 5 else torch.nn.Conv2d)(1, 4, 3, 1, 1).to(dev).eval()
            x = torch.rand(shape, device=dev, requires_grad=True)
            t0 = time.time()
            y = conv(x)
            g = torch.rand_like(y)
            torch.autograd.grad(y, x, g)
            torch.cuda.synchronize()
            first = time.time() - t0

            def fwd():
                with torch.no_grad():
                    conv(x)

            def fwdbwd():
                yy = conv(x)
                torch.autograd.grad(yy, x, g)
            tf, tb = timeit(fwd), timeit(fwdbwd)
            print("benchmark=%-5s %-22s first call %.2f s; forward %.1f us; forward + input gradient %.1f us"
                  % (bench, name, first, tf, tb))
            sys.stdout.flush()


if __name__ == "__main__":
    main()
