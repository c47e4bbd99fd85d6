#!/usr/bin/env python
"""Per-kernel micro-benchmark (GPU): every hot C-ABI entry at the BASELINE shapes, HIP-event timed on the
launch stream, reported against its algorithmic bytes.  Also times the stock ATen-HIP op and a
device copy of the same size for calibration.

    python tools/kernel_bench.py [--shape 3d|2d] [--reps 20] [--only name]
"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402


def timeit(fn, reps, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def _identity_grid(N, dims):
    """(N, d, *dims) identity sampling grid, channel 0 = x (fastest axis), align_corners=True convention."""
    lin = [torch.linspace(-1, 1, s) for s in dims]
    mesh = torch.meshgrid(*lin, indexing="ij")
    return torch.stack(list(reversed(mesh)), 0).unsqueeze(0).repeat(N, *([1] * (len(dims) + 1))).contiguous()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="3d")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--only", default=None)
    ap.add_argument("--batch", type=int, default=None)
    args = ap.parse_args()
    from advchain_amd import bands, ops
    from advchain_amd.augmentor import AdvMorph
    dev = torch.device("cuda")
    if args.shape == "3d":
        N, dims, vs = args.batch or 4, (128, 128, 64), [8, 8, 32]
    elif args.shape == "3d5":   # cfg-5's volume (rows of 80 voxels: x segments), the paired batch of its 4 volumes
        N, dims, vs = args.batch or 8, (160, 160, 80), [20, 20, 10]
    elif args.shape == "3d96":  # rows of 96 voxels (flat forward march with PW = 128)
        N, dims, vs = args.batch or 8, (96, 96, 96), [12, 12, 12]
    else:
        N, dims, vs = args.batch or 32, (256, 256), [16, 16]
    d = len(dims)
    V = 1
    for s in dims:
        V *= s
    NV = N * V
    torch.manual_seed(0)
    t = AdvMorph(spatial_dims=d, config_dict=dict(epsilon=1.5, data_size=[N, 1] + list(dims), vector_size=vs), device=dev)
    t.init_parameters()
    with torch.no_grad():
        q = t._field(1.0).contiguous()
    tabs = t._tables
    phi = ops.raw_tp_interp(ops.raw_gauss(t.param, d, pre=1, scale=1.5), tabs, d, add_identity=True, scale=1.0 / 256)
    x1 = torch.rand(N, 1, *dims, device=dev)
    x4 = torch.rand(N, 4, *dims, device=dev)
    g1, g4, gq = torch.rand_like(x1), torch.rand_like(x4), torch.rand_like(q)
    theta = (torch.eye(d, d + 1, device=dev).repeat(N, 1, 1) + 0.05 * torch.randn(N, d, d + 1, device=dev)).contiguous()
    rot = float(os.environ.get("KB_ROT", "0"))
    if rot:   # in-plane rotation (degrees) on top: the ascent steps of the bench workloads reach 15-30 degrees
        import math
        cr, sr = math.cos(math.radians(rot)), math.sin(math.radians(rot))
        R = torch.eye(d, device=dev)
        R[0, 0], R[0, 1], R[1, 0], R[1, 1] = cr, -sr, sr, cr
        theta = torch.cat([R @ theta[:, :, :d], theta[:, :, d:]], dim=2).contiguous()
    rows = []

    def add(name, fn, nbytes):
        if args.only and args.only not in name:
            return
        dt = timeit(fn, args.reps)
        rows.append((name, dt * 1e6, nbytes / dt / 1e9, nbytes / 1e6))

    add("copy %d ch (torch clone)" % d, lambda: q.clone(), 8 * d * NV)
    add("copy 1 ch (torch clone)", lambda: x1.clone(), 8 * NV)
    add("grid_sample fwd C=1", lambda: ops.raw_grid_sample_fwd(x1, q, 0, 0, True), 4 * NV * (2 + d))
    add("grid_sample fwd C=4", lambda: ops.raw_grid_sample_fwd(x4, q, 0, 0, True), 4 * NV * (8 + d))
    add("grid_sample bwd C=1 (gin+ggrid)", lambda: ops.raw_grid_sample_bwd(g1, x1, q, 0, 0, True, True, True), 4 * NV * (3 + 2 * d))
    add("grid_sample bwd C=4 (gin+ggrid)", lambda: ops.raw_grid_sample_bwd(g4, x4, q, 0, 0, True, True, True), 4 * NV * (12 + 2 * d))
    hq = ops.warp_halo(ops.grid_displacement(q), d)
    add("grid_sample bwd C=1 (gin+ggrid) halo=%d" % hq, lambda: ops.raw_grid_sample_bwd(g1, x1, q, 0, 0, True, True, True, hq), 4 * NV * (3 + 2 * d))
    add("grid_sample bwd C=4 (gin+ggrid) halo=%d" % hq, lambda: ops.raw_grid_sample_bwd(g4, x4, q, 0, 0, True, True, True, hq), 4 * NV * (12 + 2 * d))
    add("grid_sample bwd C=4 (gin only) halo=%d" % hq, lambda: ops.raw_grid_sample_bwd(g4, x4, q, 0, 0, True, True, False, hq), 4 * NV * (12 + d))
    add("grid_sample bwd C=1 (ggrid only)", lambda: ops.raw_grid_sample_bwd(g1, x1, q, 0, 0, True, False, True), 4 * NV * (2 + 2 * d))
    add("grid_sample bwd C=4 (gin only)", lambda: ops.raw_grid_sample_bwd(g4, x4, q, 0, 0, True, True, False), 4 * NV * (12 + d))
    qq = q.permute(0, *range(2, 2 + d), 1).contiguous()
    add("ATen F.grid_sample fwd C=1", lambda: F.grid_sample(x1, qq, align_corners=True), 4 * NV * (2 + d))
    xr, qr = x1.clone().requires_grad_(True), qq.clone().requires_grad_(True)

    def aten_fb():
        o = F.grid_sample(xr, qr, align_corners=True)
        torch.autograd.grad(o, (xr, qr), g1)
    add("ATen F.grid_sample fwd+bwd C=1", aten_fb, 4 * NV * (5 + 3 * d))
    add("compose_self fwd", lambda: ops.raw_compose_self_fwd(phi), 8 * d * NV)
    add("compose_self bwd", lambda: ops.raw_compose_self_bwd(gq, phi), 12 * d * NV)
    ws = ops._scatter_workspace(N, dims, dev)   # phi is the 2^-8-scaled field: well below one voxel
    add("compose_self bwd halo=1 (gather form)", lambda: ops.raw_compose_self_bwd(gq, phi, ws, False, 1), 12 * d * NV)
    add("compose_self bwd halo=-1 (exact bound)", lambda: ops.raw_compose_self_bwd(gq, phi, ws, False, -1), 12 * d * NV)
    if d == 3:   # fields of 1.5 / 3.5 voxels: exact-bound owner-computes march against the window scatter (halo 8)
        for amp in (1.5, 3.5, 6.5):
            low = torch.rand(N, d, *[max(2, s // 8) for s in dims], device=dev) * 2 - 1
            up = F.interpolate(low, size=dims, mode="trilinear", align_corners=True)
            up = up / up.abs().max()
            sc = torch.tensor([2.0 * amp / (dims[d - 1 - a] - 1) for a in range(d)], device=dev).view(1, d, 1, 1, 1)
            big = (_identity_grid(N, dims).to(dev) + up * sc).contiguous()
            hb = -int(amp + 1)
            add("compose_self bwd %.1f vox halo=%d (march)" % (amp, hb), lambda big=big, hb=hb: ops.raw_compose_self_bwd(gq, big, ws, False, hb), 12 * d * NV)
            add("compose_self bwd %.1f vox halo=8 (window)" % amp, lambda big=big: ops.raw_compose_self_bwd(gq, big, ws, False, 8), 12 * d * NV)
            add("grid_sample bwd C=1 %.1f vox halo=%d (march)" % (amp, hb), lambda big=big, hb=hb: ops.raw_grid_sample_bwd(g1, x1, big, 0, 0, True, True, True, hb), 4 * NV * (3 + 2 * d))
            add("grid_sample bwd C=1 %.1f vox halo=8 (window)" % amp, lambda big=big: ops.raw_grid_sample_bwd(g1, x1, big, 0, 0, True, True, True, 8), 4 * NV * (3 + 2 * d))
            add("grid_sample bwd C=4 %.1f vox halo=%d (march)" % (amp, hb), lambda big=big, hb=hb: ops.raw_grid_sample_bwd(g4, x4, big, 0, 0, True, True, True, hb), 4 * NV * (12 + 2 * d))
            add("grid_sample bwd C=4 %.1f vox halo=8 (window)" % amp, lambda big=big: ops.raw_grid_sample_bwd(g4, x4, big, 0, 0, True, True, True, 8), 4 * NV * (12 + 2 * d))
    if d == 2:   # fields of 1.5 / 3.5 / 6 px: exact-bound gather form (H = 2, 4) against the window scatter
        for amp in (1.5, 3.5, 6.0, 12.0, 24.0):
            low = torch.rand(N, d, *[max(2, s // 8) for s in dims], device=dev) * 2 - 1
            up = F.interpolate(low, size=dims, mode="bilinear", align_corners=True)
            up = up / up.abs().max()
            sc = torch.tensor([2.0 * amp / (dims[d - 1 - a] - 1) for a in range(d)], device=dev).view(1, d, 1, 1)
            big = (_identity_grid(N, dims).to(dev) + up * sc).contiguous()
            cand = [16] + [-h for h in (2, 4, 8, 16, 32) if amp < h and h < 4 * amp]
            for hb in cand:
                add("compose_self bwd %.1f px halo=%d" % (amp, hb), lambda big=big, hb=hb: ops.raw_compose_self_bwd(gq, big, ws, False, hb), 12 * d * NV)
                if hb == -32:      # (whole-row scatter of the squarings only)
                    continue
                add("grid_sample bwd C=4 %.1f px halo=%d" % (amp, hb), lambda big=big, hb=hb: ops.raw_grid_sample_bwd(g4, x4, big, 0, 0, True, True, True, hb), 4 * NV * (12 + 2 * d))
                add("grid_sample bwd C=1 %.1f px halo=%d" % (amp, hb), lambda big=big, hb=hb: ops.raw_grid_sample_bwd(g1, x1, big, 0, 0, True, True, True, hb), 4 * NV * (3 + 2 * d))
    add("compose_self bwd halo=2%s" % (" (gather form)" if d == 2 else ""), lambda: ops.raw_compose_self_bwd(gq, phi, ws, False, 2), 12 * d * NV)
    old = ops.TILED_SCATTER
    ops.TILED_SCATTER = False
    add("compose_self bwd (atomic path)", lambda: ops.raw_compose_self_bwd(gq, phi), 12 * d * NV)
    ops.TILED_SCATTER = old
    if d == 2:   # the whole chain of 8 squarings on the paired batch [v; -v] of a solver step, with and without the fused levels
        import ctypes
        from advchain_amd import _lib
        lib = _lib.load()
        n = 8
        s1 = ops.raw_gauss_small_pair(t.param, 1.5 * float(os.environ.get("KB_CHAIN_AMP", "4")))   # ~ the field after one ascent step
        rows_d = torch.zeros(n + 2, ops.DISP_SLOTS, device=dev)
        p0 = ops.raw_tp_interp(s1, tabs, d, add_identity=True, scale=1.0 / 2 ** n, disp_out=rows_d[0])
        NB = p0.shape[0]
        fields = torch.empty((n - 1,) + tuple(p0.shape), device=dev)
        pos = torch.empty_like(p0)
        da = _lib.dims_array(dims)

        def chain_fwd(hints, fuse):
            rows_d[1:].zero_()
            harr = None if hints is None else (ctypes.c_int32 * n)(*hints)
            _lib.check(lib.advchain_expo_chain_fwd(ops._ptr(p0), ops._ptr(fields), ops._ptr(pos), NB, d, da, n, ops._ptr(rows_d), harr,
                                                   ops._ptr(rows_d[n + 1]) if fuse else None, ops._stream()), "chain_fwd")
        chain_fwd(None, False)
        dm = ops.raw_slot_rows_max(rows_d).tolist()
        hints = [(ops._hint_bits(x) >> 8) | (ops._fine_bits(x) << 8) for x in dm[:n]]
        halos = [ops.squaring_halo(dm[m], d) for m in range(n - 1, -1, -1)]
        gp = torch.rand_like(p0)
        wsb = ops._scatter_workspace(NB, dims, dev)
        gout, scr = torch.empty_like(gp), torch.empty_like(gp)
        phis = [p0] + list(fields.unbind(0))
        bufs = [torch.empty_like(gp), torch.empty_like(gp)]

        def chain_bwd_steps():
            g = gp
            for i, ph in enumerate(reversed(phis)):
                _lib.check(lib.advchain_compose_self_bwd(ops._ptr(g), ops._ptr(ph), ops._ptr(bufs[i % 2]), ops._ptr(wsb), int(i > 0), halos[i],
                                                         NB, d, da, ops._stream()), "step")
                g = bufs[i % 2]

        def chain_bwd():
            _lib.check(lib.advchain_expo_chain_bwd(ops._ptr(gp), ops._ptr(p0), ops._ptr(fields), ops._ptr(gout), ops._ptr(scr), ops._ptr(wsb),
                                                   (ctypes.c_int32 * n)(*halos), NB, d, da, n, ops._stream()), "chain_bwd")
        tag = "disp %s" % " ".join("%.1f" % x for x in dm[:n])
        add("expo_chain fwd x8, paired (%s)" % tag, lambda: chain_fwd(hints, False), (8 * n + 4) * d * NB * V)
        add("expo_chain fwd x8, fused levels", lambda: chain_fwd(hints, True), (8 * n + 4) * d * NB * V)
        add("expo_chain bwd x8, separate launches", chain_bwd_steps, 12 * n * d * NB * V)
        add("expo_chain bwd x8, fused tail %s" % halos, chain_bwd, 12 * n * d * NB * V)
    add("gauss %d passes (d ch, pre2/post1)" % d, lambda: ops.raw_gauss(q, d, pre=2, post=1), 8 * d * NV * d)
    add("affine_warp fwd C=1", lambda: ops.affine_warp(x1, theta), 8 * NV)
    add("affine_warp fwd C=4", lambda: ops.affine_warp(x4, theta), 32 * NV)
    xa, ta = x4.clone().requires_grad_(True), theta.clone().requires_grad_(True)

    def aff_fb():
        o = ops.affine_warp(xa, ta)
        torch.autograd.grad(o, (xa, ta), g4)
    add("affine_warp fwd+bwd C=4", aff_fb, (32 + 48) * NV)
    add("tp_interp (phi0 init)", lambda: ops.raw_tp_interp(t.param, tabs, d, add_identity=True, scale=1 / 256.), 4 * d * NV)
    add("tp_adjoint (upsample bwd)", lambda: ops.raw_tp_adjoint(gq, tabs, gfull2=gq, scale=1 / 256.), 8 * d * NV)
    add("axpy", lambda: ops.raw_axpy(x1, g1, 0.5), 12 * NV)
    add("normalized_axpy", lambda: ops.normalized_axpy(x1, g1, 1.0), 20 * NV)
    from advchain_amd.common.loss import calc_segmentation_consistency
    pr = x4.clone().requires_grad_(True)
    m1 = (torch.rand_like(x1) > 0.1).float().expand(-1, 4, *([-1] * d))

    def loss_fb():
        v = calc_segmentation_consistency(pr, g4, ['mse', 'contour'], [1.0, 0.5], mask=m1)
        torch.autograd.grad(v, pr)
    add("consistency loss fwd+bwd K=4", loss_fb, 4 * NV * (4 + 4 + 1 + 4))
    if d == 2:
        # the bf16 STORAGE experiment (include/advchain_hip.h, last section): the fused loss entries called raw, fp32 against bf16
        # storage of pred / ref / R / grad_pred on the same (bf16-representable) values; bytes = what each form moves
        from advchain_amd import _lib
        lib = _lib.load()
        dm = _lib.dims_array(dims)
        mk = (torch.rand_like(x1) > 0.1).float()
        pf, rf = x4.bfloat16().float().contiguous(), g4.bfloat16().float().contiguous()
        pb, rb = pf.bfloat16().contiguous(), rf.bfloat16().contiguous()
        Rf = torch.empty(N, 6, *dims, device=dev)
        Rb = torch.empty(N, 6, *dims, device=dev, dtype=torch.bfloat16)
        gf, gb = torch.empty_like(pf), torch.empty_like(pb)
        slots = torch.zeros(4, 64, device=dev)
        st = ops._stream()
        P = lambda t: ctypes.c_void_p(t.data_ptr())

        def f32_fwd():
            _lib.check(lib.advchain_consistency_fused_fwd(P(pf), P(rf), P(mk), P(Rf), P(slots), N, 4, 2, dm, 1, 0, 1, 0, st), "fwd")

        def f32_bwd():
            _lib.check(lib.advchain_consistency_fused_bwd(P(pf), P(rf), P(Rf), P(mk), None, P(gf), 1.0, 0.5, 0.5, 0.0, 0, N, 4, 2, dm, 1, st), "bwd")

        def b16_fwd():
            _lib.check(lib.advchain_consistency_fused_fwd_bf16(P(pb), P(rb), P(mk), P(Rb), P(slots), N, 4, 2, dm, st), "fwd16")

        def b16_bwd():
            _lib.check(lib.advchain_consistency_fused_bwd_bf16(P(pb), P(rb), P(Rb), P(mk), None, P(gb), 1.0, 0.5, 0.5, N, 4, 2, dm, st), "bwd16")
        add("fused loss fwd K=4 fp32 storage", f32_fwd, 4 * NV * (4 + 4 + 1 + 6))
        add("fused loss fwd K=4 bf16 storage", b16_fwd, 2 * NV * (4 + 4 + 6) + 4 * NV)
        add("fused loss bwd K=4 fp32 storage", f32_bwd, 4 * NV * (4 + 4 + 6 + 1 + 4))
        add("fused loss bwd K=4 bf16 storage", b16_bwd, 2 * NV * (4 + 4 + 6 + 4) + 4 * NV)
    print("shape %s N=%d dims=%s" % (args.shape, N, dims))
    print("%-40s %10s %10s %10s" % ("kernel", "us", "GB/s(alg)", "MB(alg)"))
    for r in rows:
        print("%-40s %10.1f %10.1f %10.1f" % r)


if __name__ == "__main__":
    main()
