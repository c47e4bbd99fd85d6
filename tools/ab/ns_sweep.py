#!/usr/bin/env python
"""North-star pair (grid_sample fwd + bwd, 4x1x128x128x64, fresh AdvMorph field): event timing of each kernel, for tuning
sweeps over ADVCHAIN_FWD_MARCH_ZC / ADVCHAIN_MARCH_ZC (knobs are read once per process: one run per setting)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from advchain_amd import ops  # noqa: E402
from advchain_amd.augmentor import AdvMorph  # noqa: E402

dev = torch.device("cuda")
ds = [4, 1, 128, 128, 64]
torch.manual_seed(0)
t = AdvMorph(spatial_dims=3, config_dict=dict(epsilon=1.5, data_size=ds, vector_size=[8, 8, 32]), device=dev)
t.init_parameters()
with torch.no_grad():
    q = t._field(1.0).contiguous()
x, go = torch.rand(*ds, device=dev), torch.rand(*ds, device=dev)
halo = ops.warp_halo(ops.grid_displacement(q), 3)


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


f = timeit(lambda: ops.raw_grid_sample_fwd(x, q, 0, 0, True))
b = timeit(lambda: ops.raw_grid_sample_bwd(go, x, q, 0, 0, True, True, True, halo))
print("FWD_ZC=%s BWD_ZC=%s  fwd %.2f us  bwd %.2f us  pair %.2f us -> %.3f of 8 TB/s" % (
    os.environ.get("ADVCHAIN_FWD_MARCH_ZC", "-"), os.environ.get("ADVCHAIN_MARCH_ZC", "-"), f, b, f + b, 234881024 / (f + b) / 8e6))
