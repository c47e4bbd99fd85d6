#!/usr/bin/env python
"""f1 experiment: the scaling-and-squaring chain with pairs of squarings fused into one launch (ADVCHAIN_FUSE2=2) against
one launch per squaring.  Run once per setting; --save / --check compare the fields bit for bit.

    python tools/ab/fuse2_ab.py [--save f.pt | --check f.pt]
"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--save", default=None)
    ap.add_argument("--check", default=None)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--amp", type=float, default=1.5)
    args = ap.parse_args()
    from advchain_amd import _lib, bands, ops
    from advchain_amd.augmentor import AdvMorph
    dev = torch.device("cuda")
    N, dims, vs = args.batch, (128, 128, 64), [8, 8, 32]
    torch.manual_seed(0)
    t = AdvMorph(spatial_dims=3, config_dict=dict(epsilon=args.amp, data_size=[N, 1] + list(dims), vector_size=vs), device=dev)
    t.init_parameters()
    tabs = bands.upsample_tables(vs, list(dims), dev)
    n = 8
    phi0 = ops.raw_tp_interp(ops.raw_gauss(t.param, 3, pre=1, scale=args.amp), tabs, 3, add_identity=True, scale=1.0 / 2 ** n)
    fields = torch.empty((n - 1,) + tuple(phi0.shape), device=dev)
    pos = torch.empty_like(phi0)
    disp = torch.zeros(n + 1, ops.DISP_SLOTS, device=dev)
    lib = _lib.load()

    def run():
        _lib.check(lib.advchain_expo_chain_fwd(ops._ptr(phi0), ops._ptr(fields), ops._ptr(pos), N, 3, _lib.dims_array(dims), n,
                                               ops._ptr(disp), None, None, ops._stream()), "chain")
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    print("FUSE2=%s  batch %d  expo_chain_fwd (8 squarings): %.1f us per chain; displacement of phi_1..pos (voxels): %s"
          % (os.environ.get("ADVCHAIN_FUSE2", "0"), N, e0.elapsed_time(e1) * 1e3 / reps,
             ["%.3f" % float(v) for v in disp.max(dim=1).values[1:]]))
    res = dict(fields=fields.cpu(), pos=pos.cpu(), disp=disp.max(dim=1).values.cpu())
    if args.save:
        torch.save(res, args.save)
    if args.check:
        ref = torch.load(args.check)
        for k in res:
            print("  %-6s equal=%s max|diff| %.3e" % (k, torch.equal(res[k], ref[k]), float((res[k] - ref[k]).abs().max())))
        print("  per field phi_1..phi_7 max|diff|:", ["%.1e" % float((res["fields"][m] - ref["fields"][m]).abs().max()) for m in range(n - 1)])


if __name__ == "__main__":
    main()
