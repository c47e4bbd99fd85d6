import sys, os
sys.path.insert(0, "/root/repo")
import torch
from tests.test_ops_gpu import _smooth_field, rand
from advchain_amd import ops
for dims, amp in (((24, 21, 44), 0.4), ((24, 21, 44), 2.6), ((16, 32, 64), 0.6)):
    phi = _smooth_field(dims, amp, 67).cuda()
    for C in (1, 4):
        x = rand((2, C) + dims, 68 + C).cuda()
        for clamp in (True, False):
            outs = {h: ops.raw_grid_sample_fwd(x, phi, 0, 0, clamp, disp_hint=h) for h in (None, 0.5, 4.5, 9.0)}
            for h in (0.5, 4.5, 9.0):
                dd = (outs[h] - outs[None]).abs()
                print(dims, amp, "C", C, "clamp", clamp, "hint", h, "ndiff", int((dd > 0).sum()), "max", float(dd.max()))
    comps = {h: ops.raw_compose_self_fwd(phi, disp_hint=h) for h in (None, 0.5, 4.5, 9.0)}
    for h in (0.5, 4.5, 9.0):
        dd = (comps[h] - comps[None]).abs()
        print(dims, amp, "self hint", h, "ndiff", int((dd > 0).sum()), "max", float(dd.max()))
