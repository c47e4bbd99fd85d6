#!/usr/bin/env python
"""Displacement of phi_0..phi_n (pixels) of every scaling-and-squaring chain of one cfg-2 solver call, as the backward
reads them back -- what the kernel-selection hints of the NEXT chain are made of."""
import os, sys, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from advchain_amd import ops
wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg2"]
dev = torch.device("cuda")
solver = bench.build_solver(wl, dev)
torch.manual_seed(0)
data = torch.rand(wl["batch"], 1, *wl["dims"], device=dev)
model = bench.make_model(len(wl["dims"])).to(dev)
orig = ops._HintCache.__setitem__
def rec(self, k, v):
    if self is ops._CHAIN_HINTS:
        print(" ".join("%6.2f" % x for x in v))
    orig(self, k, v)
ops._HintCache.__setitem__ = rec
for call in range(2):
    print("call", call)
    with contextlib.redirect_stdout(io.StringIO()) as buf:
        pass
    solver.adversarial_training(data=data, model=model, **bench.solver_kwargs(wl, dev))
print(ops.FUSE_STATS)
