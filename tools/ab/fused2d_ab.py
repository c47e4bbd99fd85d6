#!/usr/bin/env python
"""The 2D scaling-and-squaring chain (forward / backward) with and without the fused early squarings, on the field of a
cfg-2 AdvMorph (paired batch of 64 x 2 x 256 x 256): us per chain call, events on the launch stream.

    python tools/ab/fused2d_ab.py [--batch 32] [--amp 1.5] [--k 4]
"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--amp", type=float, default=1.5)
    ap.add_argument("--dims", type=int, nargs=2, default=[256, 256])
    ap.add_argument("--reps", type=int, default=30)
    args = ap.parse_args()
    from advchain_amd import _lib, bands, ops
    dev = torch.device("cuda")
    dims = tuple(args.dims)
    vs = [dims[0] // 16, dims[1] // 16]
    N = args.batch
    torch.manual_seed(0)
    v = torch.rand(N, 2, *vs, device=dev) * 2 - 1
    v = v / v.reshape(N, -1).norm(dim=1).view(N, 1, 1, 1)
    tabs = bands.upsample_tables(vs, list(dims), dev)
    n = 8
    s1 = ops.raw_gauss_small_pair(v, args.amp)
    disp = torch.zeros(n + 2, ops.DISP_SLOTS, device=dev)
    phi0 = ops.raw_tp_interp(s1, tabs, 2, add_identity=True, scale=1.0 / 2 ** n, disp_out=disp[0])
    fields = torch.empty((n - 1,) + tuple(phi0.shape), device=dev)
    pos = torch.empty_like(phi0)
    lib = _lib.load()
    NB = phi0.shape[0]

    def fwd(hints, fuse):
        disp[1:].zero_()
        harr = None if hints is None else (ctypes.c_int32 * n)(*hints)
        _lib.check(lib.advchain_expo_chain_fwd(ops._ptr(phi0), ops._ptr(fields), ops._ptr(pos), NB, 2, _lib.dims_array(dims), n,
                                               ops._ptr(disp), harr, ops._ptr(disp[n + 1]) if fuse else None, ops._stream()),
                   "chain")
    fwd(None, False)
    dm = ops.raw_slot_rows_max(disp).tolist()
    hints = [(ops._hint_bits(x) >> 8) | (ops._fine_bits(x) << 8) for x in dm[:n]]
    print("displacement of phi_0..phi_%d (px): %s" % (n - 1, " ".join("%.2f" % x for x in dm[:n])), " pos %.2f" % dm[n])
    print("hints", hints)
    ref = (fields.clone(), pos.clone())

    def timeit(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / args.reps
    t_plain = timeit(lambda: fwd(hints, False))
    t_fused = timeit(lambda: fwd(hints, True))
    fwd(hints, True)
    torch.cuda.synchronize()
    same = torch.equal(fields, ref[0]) and torch.equal(pos, ref[1])
    print("expo_chain_fwd  batch %d x 2 x %d x %d: one launch per squaring %.1f us, fused leading squarings %.1f us; flag %.0f; bit-identical %s"
          % (NB, dims[0], dims[1], t_plain, t_fused, float(disp[n + 1, 0]), same))
    for k in (2, 3, 4, 5):
        hk = [1 | (1 << 8)] * k + [9] * (n - k)
        if all((h & 0xff) == 1 for h in hints[:k]):
            print("   k = %d: %.1f us" % (k, timeit(lambda: fwd(hk, True))))
    # ---- backward: the chain entry (fuses the trailing exact sub-pixel steps) against n separate calls
    fwd(None, False)
    halos = [ops.squaring_halo(dm[m], 2) for m in range(n - 1, -1, -1)]
    gpos = torch.rand_like(phi0)
    ws = ops._scatter_workspace(NB, dims, dev)
    out, scratch = torch.empty_like(gpos), torch.empty_like(gpos)
    harr = (ctypes.c_int32 * n)(*halos)

    def bwd_chain():
        _lib.check(lib.advchain_expo_chain_bwd(ops._ptr(gpos), ops._ptr(phi0), ops._ptr(fields), ops._ptr(out), ops._ptr(scratch),
                                               ops._ptr(ws), harr, NB, 2, _lib.dims_array(dims), n, ops._stream()), "bwd")
    phis = [phi0] + list(fields.unbind(0))
    bufs = [torch.empty_like(gpos), torch.empty_like(gpos)]

    def bwd_steps():
        g = gpos
        for i, phi in enumerate(reversed(phis)):
            o = bufs[i % 2]
            _lib.check(lib.advchain_compose_self_bwd(ops._ptr(g), ops._ptr(phi), ops._ptr(o), ops._ptr(ws), int(i > 0), halos[i], NB, 2,
                                                     _lib.dims_array(dims), ops._stream()), "step")
            g = o
        return g
    t_steps, t_chain = timeit(bwd_steps), timeit(bwd_chain)
    ref_g = bwd_steps().clone()
    bwd_chain()
    torch.cuda.synchronize()
    print("expo_chain_bwd  halos %s: %d separate launches %.1f us, chain entry (fused tail) %.1f us; max |diff| %.3e of %.3e"
          % (halos, n, t_steps, t_chain, float((out - ref_g).abs().max()), float(ref_g.abs().max())))


if __name__ == "__main__":
    main()
