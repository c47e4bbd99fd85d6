"""The fused early squarings of the 2D scaling-and-squaring chain (advchain_amd/csrc/expo_fused2d.hip; reference loop
adv_morph.py:116-146) against the one-launch-per-squaring chain: bit for bit -- fields, positions, displacement rows --
whether the sub-pixel premise holds (one fused launch, the repeat launch behind it returns at once), fails everywhere (the fused
kernel only raises its flag) or fails for ONE window of one image (the ordinary launches redo every squaring)."""
import ctypes

import pytest
import torch

from tests.helpers import rand

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _chain(phi0, n, hints, fuse):
    from advchain_amd import _lib, ops
    N, d = phi0.shape[0], phi0.dim() - 2
    dims = tuple(phi0.shape[2:])
    lib = _lib.load()
    rows = torch.zeros(n + 2, ops.DISP_SLOTS, device=DEV)
    fields = torch.full((n - 1,) + tuple(phi0.shape), float("nan"), device=DEV)
    pos = torch.empty_like(phi0)
    # (bits 8.. of a hint: the displacement estimate in 1/1024 pixel; 1 = "known and tiny" -- the tests decide k by how many
    # leading hints say so)
    harr = None if hints is None else (ctypes.c_int32 * n)(*[h | ((1 << 8) if h == 1 else 0) for h in hints])
    _lib.check(lib.advchain_expo_chain_fwd(ops._ptr(phi0), ops._ptr(fields), ops._ptr(pos), N, d, _lib.dims_array(dims), n,
                                           ops._ptr(rows), harr, ops._ptr(rows[n + 1]) if fuse else None, ops._stream()),
               "expo_chain_fwd")
    torch.cuda.synchronize()
    return fields, pos, rows[:n + 1].max(1).values, float(rows[n + 1, 0])


def _phi0(N, dims, amp_px, seed):
    """identity + a smooth displacement of at most amp_px pixels"""
    from oracle import advchain_oracle as O
    import torch.nn.functional as F
    low = rand((N, 2, 5, 5), seed)
    u = F.interpolate(low, size=dims, mode="bilinear", align_corners=True)
    u = u / u.abs().max()
    scale = torch.tensor([2.0 / (dims[1] - 1), 2.0 / (dims[0] - 1)]).view(1, 2, 1, 1)      # channel 0 = x <-> last axis
    return (O.identity_grid(N, dims) + amp_px * u * scale).contiguous().to(DEV)


@pytest.mark.parametrize("dims", [(256, 256), (192, 192), (40, 64), (100, 128), (37, 320), (24, 512)])
@pytest.mark.parametrize("k", [2, 3, 4, 5])
def test_fused_levels_equal_the_per_squaring_launches(dims, k):
    n = 8
    phi0 = _phi0(max(3, -(-256 // max(1, dims[0] // 16))), dims, 0.9 / 2 ** (k - 1) * 0.95, 11 + k)   # phi_{k-1} just below one pixel; >= 256 windows

    hints = [1] * k + [0] * (n - k)
    ref = _chain(phi0, n, None, False)
    out = _chain(phi0, n, hints, True)
    assert out[3] == 0.0, "the fused kernel raised its flag on a sub-pixel field"
    assert torch.equal(out[1], ref[1]) and torch.equal(out[2], ref[2])
    for m in range(n - 1):
        assert torch.equal(out[0][m], ref[0][m]), (dims, k, m)
    # the repeat launch really returned at once: poison the first k fields, run with hints again -> still the fused values
    # (a fused launch that silently did nothing would leave NaN here, since the repeat launch does not run either)
    assert not torch.isnan(out[0][:k]).any()


@pytest.mark.parametrize("dims", [(256, 256), (100, 128), (37, 320)])
def test_fused_levels_above_one_pixel_ride_on_a_wider_halo(dims):
    """Hints with the measured displacements (bits 8..: 1/1024 px): the squaring whose input moves 1 .. 1.6 px joins the fused
    launch with a row halo of 2 (the window carries the sum of the halos, at most 5 rows either side) -- same bits, flag
    down; a hint that promises less than the field does (halo 2 for an input that moves 2.4 px) raises a deficit of one
    level and the repeat launch redoes it."""
    import ctypes
    from advchain_amd import _lib, ops
    n = 8
    N = max(3, -(-256 // max(1, dims[0] // 16)))
    phi0 = _phi0(N, dims, 0.19, 17)                  # phi_j moves ~0.19 * 2^j px: 0.19 0.38 0.76 1.5 3 ...
    ref = _chain(phi0, n, None, False)
    dm = ref[2].tolist()
    assert dm[2] < 0.8 and 1.0 < dm[3] < 1.6, dm[:5]
    lib = _lib.load()

    def run(hint_disp):
        rows = torch.zeros(n + 2, ops.DISP_SLOTS, device=DEV)
        fields = torch.full((n - 1,) + tuple(phi0.shape), float("nan"), device=DEV)
        pos = torch.empty_like(phi0)
        harr = (ctypes.c_int32 * n)(*[(ops._fine_bits(x) << 8) if x is not None else 0 for x in hint_disp])
        _lib.check(lib.advchain_expo_chain_fwd(ops._ptr(phi0), ops._ptr(fields), ops._ptr(pos), phi0.shape[0], 2, _lib.dims_array(dims), n,
                                               ops._ptr(rows), harr, ops._ptr(rows[n + 1]), ops._stream()), "expo_chain_fwd")
        torch.cuda.synchronize()
        return fields, pos, rows[:n + 1].max(1).values, float(rows[n + 1, 0])
    out = run(dm[:4] + [None] * 4)                   # levels 1..4 fused: halos 1, 1, 1, 2
    assert out[3] == 0.0
    lying = run(dm[:3] + [dm[3] / 2.0, None, None, None, None]) if dm[3] * 1.25 / 2 < 1.0 else None
    for res in (out, lying):
        if res is None:
            continue
        assert torch.equal(res[1], ref[1]) and torch.equal(res[2], ref[2])
        for m in range(n - 1):
            assert torch.equal(res[0][m], ref[0][m]), (dims, m)
    if lying is not None:
        assert lying[3] == 1.0                       # the level promised as sub-pixel was not: one level redone


def test_partial_deficit_repeats_only_the_missing_levels():
    """Hints promise 4 sub-pixel levels, the field allows 2 (phi_2 moves more than a pixel): the fused kernel stops after
    level 2 everywhere it must, records a deficit of 2, and the gated launches of levels 3 and 4 run -- same bits."""
    n, dims = 8, (128, 256)
    phi0 = _phi0(32, dims, 0.4, 9)                   # phi_1 ~0.8 px, phi_2 ~1.6 px
    ref = _chain(phi0, n, None, False)
    assert ref[2][1] < 0.999 and ref[2][2] > 1.0, ref[2][:4]
    out = _chain(phi0, n, [1, 1, 1, 1, 0, 0, 0, 0], True)
    assert out[3] == 2.0
    assert torch.equal(out[1], ref[1]) and torch.equal(out[2], ref[2])
    for m in range(n - 1):
        assert torch.equal(out[0][m], ref[0][m]), m


@pytest.mark.parametrize("case", ["all", "one_window", "nan"])
def test_fused_premise_violated_falls_back(case):
    """Hints that are too optimistic: the kernel's own check refuses, raises the flag, and the gated ordinary launches
    produce the fields -- results never depend on the hints."""
    n, dims = 8, (128, 256)
    phi0 = _phi0(32, dims, 0.05, 3)
    if case == "all":
        phi0 = _phi0(32, dims, 3.0, 3)
    elif case == "one_window":      # a 2-pixel bump in a few rows of one image
        phi0[2, 0, 70:74, 100:140] += 2.0 * 2.0 / (dims[1] - 1)
    else:
        phi0[1, 1, 5, 7] = float("nan")
    hints = [1, 1, 1, 1, 0, 0, 0, 0]
    ref = _chain(phi0, n, None, False)
    out = _chain(phi0, n, hints, True)
    assert out[3] == 4.0          # the deficit: a window could not do any of the 4 levels
    for m in range(n - 1):
        assert torch.equal(torch.nan_to_num(out[0][m], nan=7.0), torch.nan_to_num(ref[0][m], nan=7.0)), (case, m)
    assert torch.equal(torch.nan_to_num(out[1], nan=7.0), torch.nan_to_num(ref[1], nan=7.0))
    assert torch.equal(torch.nan_to_num(out[2], nan=7.0), torch.nan_to_num(ref[2], nan=7.0))


@pytest.mark.parametrize("N,dims", [(32, (48, 100)), (2, (128, 256))])
def test_unsupported_shapes_take_the_ordinary_launches(N, dims):
    """Rows that are not a multiple of 64 pixels, or fewer than 256 windows in all (3D fields likewise): the chain ignores
    the flag and runs its ordinary launches."""
    n = 5
    phi0 = _phi0(N, dims, 0.05, 5)
    ref = _chain(phi0, n, None, False)
    out = _chain(phi0, n, [1] * n, True)
    assert out[3] == 0.0 and torch.equal(out[1], ref[1])
    for m in range(n - 1):
        assert torch.equal(out[0][m], ref[0][m])


def test_demons_field_uses_the_fused_chain_and_matches():
    """Through the product operator: the second evaluation of a field of one shape has hints and fuses; same bits as with
    fusing switched off (ops.FUSE_2D)."""
    from advchain_amd import bands, ops
    dims, vs, N = (256, 256), [16, 16], 8        # (paired: 16 fields x 16 windows -- the fused forward wants 256 workgroups)
    tabs = bands.upsample_tables(vs, list(dims), DEV)
    v = rand((N, 2) + tuple(vs), 21).to(DEV)
    v = v / v.reshape(N, -1).norm(dim=1).view(N, 1, 1, 1)
    outs, levels = {}, {}
    for fuse in (False, True):
        ops.FUSE_2D = fuse
        ops.COUNT_FUSED = True
        ops._CHAIN_HINTS.clear()
        ops.FUSE_STATS["fused_levels"] = 0
        try:
            res = []
            for rep in range(2):
                vv = v.clone().requires_grad_(True)
                qp, qm = ops.demons_field_pair(vv, 1.5, tabs, False)
                (qp.sum() + 2 * qm.sum()).backward()          # (the backward records the hints the next forward uses)
                res.append((qp.detach().clone(), qm.detach().clone(), vv.grad.clone()))
            outs[fuse] = res
            levels[fuse] = ops.FUSE_STATS["fused_levels"]
        finally:
            ops.FUSE_2D = True
            ops.COUNT_FUSED = False
    key = [k for k in ops._CHAIN_HINTS][0]
    assert ops._CHAIN_HINTS[key][-1] == 0.0          # the fuse flag rode back with the displacement rows: never raised
    # the second evaluation really took the fused launch (advchain_expo_chain_fused_levels: the library's own answer for this
    # shape and these hints), the first -- no hints yet -- and the FUSE_2D = False run did not
    assert levels[True] >= 2 and levels[False] == 0, levels
    for a, b in zip(outs[False], outs[True]):
        for x, y in zip(a, b):
            assert torch.equal(x, y)
    # ... and the fused fields are the ORACLE's (pinned on the reference, tests/test_oracle_golden.py): DemonsCompose of
    # +-1.5 v (adv_morph.py:454-491), un-clamped product field against the clamped one where it is inside
    from oracle import advchain_oracle as O
    for sign, q in ((1.0, outs[True][1][0]), (-1.0, outs[True][1][1])):
        want = O.demons_compose(sign * 1.5 * v.cpu(), dims, final_clamp=True)
        assert float((torch.clamp(q, -1, 1).cpu() - want).abs().max()) < 2e-5


@pytest.mark.parametrize("dims", [(256, 256), (192, 192), (40, 64), (100, 128), (37, 320), (16, 24)])
@pytest.mark.parametrize("k", [2, 3, 4])
def test_fused_backward_levels_equal_the_per_squaring_launches(dims, k):
    """advchain_expo_chain_bwd with exact sub-pixel bounds on its last k steps (one fused launch, adjoint_fused2d.hip) against
    n calls of advchain_compose_self_bwd with the same bounds: the gradient w.r.t. phi_0, bit for bit."""
    from advchain_amd import _lib, ops
    n, N = 7, 3
    phi0 = _phi0(N, dims, 0.9 / 2 ** (k - 1) * 0.95, 31 + k)
    fields, pos, dm, _ = _chain(phi0, n, None, False)
    dm = dm.tolist()
    halos = [ops.squaring_halo(dm[m], 2) for m in range(n - 1, -1, -1)]
    assert halos[-k:] == [-1] * k, halos                  # phi_0..phi_{k-1}: exact bound below one pixel
    if k < 4:
        assert halos[-k - 1] != -1, halos
    gpos = rand((N, 2) + tuple(dims), 41).to(DEV)
    ws = ops._scatter_workspace(N, dims, DEV)
    phis = [phi0] + list(fields.unbind(0))
    g = gpos
    for i, phi in enumerate(reversed(phis)):
        g = ops.raw_compose_self_bwd(g, phi, ws, chain=i > 0, halo=halos[i])
    out, scratch = torch.full_like(gpos, float("nan")), torch.full_like(gpos, float("nan"))
    lib = _lib.load()
    _lib.check(lib.advchain_expo_chain_bwd(ops._ptr(gpos), ops._ptr(phi0), ops._ptr(fields), ops._ptr(out), ops._ptr(scratch),
                                           ops._ptr(ws), (ctypes.c_int32 * n)(*halos), N, 2, _lib.dims_array(dims), n,
                                           ops._stream()), "expo_chain_bwd")
    torch.cuda.synchronize()
    if all(h < 0 for h in halos):
        assert torch.equal(out, g), float((out - g).abs().max())
    else:       # (a window-scatter step in front: float atomics, compared to rounding)
        assert float((out - g).abs().max()) <= 1e-5 * max(1.0, float(g.abs().max()))


@pytest.mark.parametrize("dims,vs,N", [((256, 256), [16, 16], 8), ((192, 192), [12, 12], 4), ((64, 96), [4, 6], 2), ((37, 50), [3, 4], 3)])
def test_composite_entries_equal_the_separate_calls(dims, vs, N):
    """advchain_demons_compose_pair_fwd / _bwd (one C call per DemonsCompose direction, csrc/demons_compose.cpp) against the
    five / six separate calls they sequence (ops.COMPOSITE = False): both fields and the velocity gradient, bit for bit;
    shapes the fast kernels do not take (rows that are not a multiple of 4) fall back without enqueueing anything."""
    from advchain_amd import bands, ops
    tabs = bands.upsample_tables(vs, list(dims), DEV)
    v = rand((N, 2) + tuple(vs), 51).to(DEV)
    v = v / v.reshape(N, -1).norm(dim=1).view(N, 1, 1, 1)
    outs = {}
    for comp in (False, True):
        ops.COMPOSITE = comp
        ops._CHAIN_HINTS.clear()
        try:
            res = []
            for rep in range(2):                  # (the second evaluation has hints: fused squarings inside the chain call)
                vv = v.clone().requires_grad_(True)
                qp, qm = ops.demons_field_pair(vv, 1.5, tabs, False)
                (qp * qp).sum().backward(retain_graph=True)
                g1 = vv.grad.clone()
                (qp.sum() + 2 * (qm * qm).sum()).backward()
                res.append((qp.detach().clone(), qm.detach().clone(), g1, vv.grad.clone()))
            outs[comp] = res
        finally:
            ops.COMPOSITE = True
    for a, b in zip(outs[False], outs[True]):
        for x, y in zip(a, b):
            assert torch.equal(x, y)
