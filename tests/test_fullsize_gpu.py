"""Full BASELINE sizes (cfg-2: 32x1x256x256, cfg-3: 4x1x128x128x64, cfg-5: 4x1x160x160x80) on the GPU, checked through size-independent
properties, since the CPU oracle needs minutes there: adjointness <A x, y> = <x, A^T y> of every forward/backward
kernel pair, linearity, identity warps, agreement of the two independent scatter implementations, bitwise
run-to-run determinism of the fixed-point scatter (3D), and one whole solver call (finite, ascent does not lower the
loss, parameters obey their constraints)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")

SHAPES = {"cfg2": dict(N=32, dims=(256, 256), vs=[16, 16]), "cfg3": dict(N=4, dims=(128, 128, 64), vs=[8, 8, 32]),
          # cfg-5's volume: rows of 80 voxels -- the flat forward march, the A / B-wave adjoint, the flat march scatter
          "cfg5": dict(N=4, dims=(160, 160, 80), vs=[20, 20, 10])}


def _morph(shape, eps=1.5, seed=0):
    from advchain_amd.augmentor import AdvMorph
    s = SHAPES[shape]
    torch.manual_seed(seed)
    t = AdvMorph(spatial_dims=len(s["dims"]), config_dict=dict(epsilon=eps, data_size=[s["N"], 1] + list(s["dims"]),
                                                               vector_size=s["vs"]), device=DEV)
    t.init_parameters()
    return t, s


def dot(a, b):
    return float((a.double() * b.double()).sum())


@pytest.mark.parametrize("shape", ["cfg2", "cfg3", "cfg5"])
def test_warp_adjoint_linearity_and_paths(shape):
    from advchain_amd import ops
    t, s = _morph(shape)
    with torch.no_grad():
        q = t._field(1.0).contiguous()
    for C in (1, 4):
        x = torch.rand(s["N"], C, *s["dims"], device=DEV)
        y = torch.rand(s["N"], C, *s["dims"], device=DEV)
        Ax = ops.raw_grid_sample_fwd(x, q, 0, 0, True)
        # linear in the image
        Ax2 = ops.raw_grid_sample_fwd(2.5 * x + y, q, 0, 0, True)
        Ay = ops.raw_grid_sample_fwd(y, q, 0, 0, True)
        assert float((Ax2 - (2.5 * Ax + Ay)).abs().max()) < 2e-5
        # adjoint: <A x, y> == <x, A^T y>
        ATy, gq = ops.raw_grid_sample_bwd(y, x, q, 0, 0, True, True, True)
        lhs, rhs = dot(Ax, y), dot(x, ATy)
        assert abs(lhs - rhs) < 1e-5 * abs(lhs), (lhs, rhs)
        # run-to-run: the 3D tiled (fixed-point) scatter is bitwise deterministic; the 2D window scatter flushes with
        # float atomics (order-dependent in the last bits)
        ATy_again, _ = ops.raw_grid_sample_bwd(y, x, q, 0, 0, True, True, False)
        if len(s["dims"]) == 3:
            assert torch.equal(ATy, ATy_again)
        else:
            assert float((ATy - ATy_again).abs().max()) < 1e-5 * float(ATy.abs().max())
        old = ops.TILED_SCATTER
        ops.TILED_SCATTER = False
        try:
            ATy_atomic, gq_atomic = ops.raw_grid_sample_bwd(y, x, q, 0, 0, True, True, True)
        finally:
            ops.TILED_SCATTER = old
        scale = float(ATy_atomic.abs().max())
        assert float((ATy - ATy_atomic).abs().max()) < 2e-5 * scale
        assert float((gq - gq_atomic).abs().max()) < 1e-4 * float(gq_atomic.abs().max())


@pytest.mark.parametrize("shape", ["cfg2", "cfg3", "cfg5"])
def test_identity_field_is_identity_warp(shape):
    from advchain_amd import ops
    t, s = _morph(shape)
    zero = torch.zeros_like(t.param)
    q = ops.demons_field(zero, 1.0, t._tables, len(s["dims"]) == 3)
    x = torch.rand(s["N"], 2, *s["dims"], device=DEV)
    out = ops.raw_grid_sample_fwd(x, q, 0, 0, True)
    # rounding doubles in each of the 8 squarings (2^8 * 6e-8 = 1.5e-5 normalised = 2e-3 px at S=256): on
    # white-noise data (|neighbour difference| ~ 1) the identity warp is exact only to that level
    assert float((out - x).abs().max()) < 1e-2
    assert float((out - x).abs().mean()) < 1e-3


@pytest.mark.parametrize("shape", ["cfg2", "cfg3", "cfg5"])
def test_compose_self_jvp_matches_vjp(shape):
    """<J dphi, g> == <dphi, J^T g> with J dphi from a central difference of the forward kernel."""
    from advchain_amd import ops
    t, s = _morph(shape)
    d = len(s["dims"])
    phi = ops.raw_tp_interp(ops.raw_gauss(t.param, d, pre=1, scale=1.5), t._tables, d, add_identity=True, scale=1 / 16.)
    # smooth direction, small enough that (almost) no sample crosses a cell boundary inside [-h, h]
    dphi = ops.raw_gauss(torch.randn_like(phi), d)
    g = torch.randn_like(phi)
    h = 1e-4
    jv = (ops.raw_compose_self_fwd(phi + h * dphi) - ops.raw_compose_self_fwd(phi - h * dphi)) / (2 * h)
    vj = ops.raw_compose_self_bwd(g, phi)
    lhs, rhs = dot(jv, g), dot(dphi, vj)
    assert abs(lhs - rhs) < 3e-2 * max(abs(lhs), abs(rhs)) + 1.0, (lhs, rhs)   # fp32 finite difference
    vj2 = ops.raw_compose_self_bwd(g, phi)
    if d == 3:
        assert torch.equal(vj, vj2)                                  # fixed-point tiles: deterministic
    else:
        assert float((vj - vj2).abs().max()) < 1e-5 * float(vj.abs().max())   # window scatter: float-atomic flush


@pytest.mark.parametrize("shape", ["cfg2", "cfg3"])
def test_linear_kernels_are_adjoint_pairs(shape):
    from advchain_amd import ops
    t, s = _morph(shape)
    d = len(s["dims"])
    N = s["N"]
    # separable Gaussian is self-adjoint
    a = torch.randn(N, d, *s["dims"], device=DEV)
    b = torch.randn(N, d, *s["dims"], device=DEV)
    lhs, rhs = dot(ops.raw_gauss(a, d), b), dot(a, ops.raw_gauss(b, d))
    assert abs(lhs - rhs) < 1e-5 * abs(lhs) + 1e-3
    # upsample and its adjoint
    v = torch.randn_like(t.param)
    up = ops.raw_tp_interp(v, t._tables, d)
    adj = ops.raw_tp_adjoint(b, t._tables)
    lhs, rhs = dot(up, b), dot(v, adj)
    assert abs(lhs - rhs) < 1e-4 * max(1.0, abs(lhs))
    # affine warp: <A x, y> == <x, A^T y>
    theta = (torch.eye(d, d + 1, device=DEV).repeat(N, 1, 1) + 0.05 * torch.randn(N, d, d + 1, device=DEV)).contiguous()
    x = torch.rand(N, 4, *s["dims"], device=DEV).requires_grad_(True)
    y = torch.rand(N, 4, *s["dims"], device=DEV)
    Ax = ops.affine_warp(x, theta)
    (gx,) = torch.autograd.grad(Ax, x, y)
    lhs, rhs = dot(Ax.detach(), y), dot(x.detach(), gx)
    assert abs(lhs - rhs) < 1e-5 * abs(lhs)


@pytest.mark.parametrize("shape", ["cfg3", "cfg5"])
@pytest.mark.parametrize("amp,bound", [(1.5, 2), (3.5, 4)])
def test_march_scatter_agrees_with_the_window_scatter_at_full_size(shape, amp, bound):
    """The self-composition backward on a smooth field of 1.5 / 3.5 voxels at the full 3D volumes: the owner-computes march
    scatter with the exact bound (LDS int32 accumulators; lane <-> flat sample on cfg-5's rows of 80) against the
    source-tiled window scatter (LDS windows flushed with float atomics) -- two independent implementations of the same
    sum -- plus the march scatter's bitwise run-to-run determinism and the forward kernels' agreement under any hint."""
    import torch.nn.functional as F
    from advchain_amd import ops
    s = SHAPES[shape]
    N, dims = s["N"], s["dims"]
    g = torch.Generator(device="cpu").manual_seed(5)
    low = torch.rand(N, 3, *[max(2, v // 8) for v in dims], generator=g) * 2 - 1
    up = F.interpolate(low, size=dims, mode="trilinear", align_corners=True)
    up = (up / up.abs().max()).to(DEV)
    lin = [torch.linspace(-1, 1, v, device=DEV) for v in dims]
    mesh = torch.meshgrid(*lin, indexing="ij")
    ident = torch.stack(list(reversed(mesh)), 0).unsqueeze(0)
    sc = torch.tensor([2.0 * amp / (dims[2 - a] - 1) for a in range(3)], device=DEV).view(1, 3, 1, 1, 1)
    phi = (ident + up * sc).contiguous()
    assert bound - 1 <= float(ops.raw_max_displacement(phi).item()) < bound - 0.001
    gq = torch.randn(N, 3, *dims, device=DEV)
    ws = ops._scatter_workspace(N, dims, DEV)
    a = ops.raw_compose_self_bwd(gq, phi, ws, False, -bound)
    assert torch.equal(a, ops.raw_compose_self_bwd(gq, phi, ws, False, -bound))
    b = ops.raw_compose_self_bwd(gq, phi, ws, False, 8)
    assert float((a - b).abs().max()) < 3e-5 * float(b.abs().max())
    outs = [ops.raw_compose_self_fwd(phi, disp_hint=h) for h in (None, 0.5, 2.5, 9.0)]
    for o in outs[1:]:
        assert torch.equal(o, outs[0])


@pytest.mark.parametrize("name", ["cfg1", "cfg2", "cfg3", "cfg4", "cfg5"])
def test_whole_solver_call_at_full_size(name):
    """One adversarial_training call of every BASELINE config at its full per-GPU size (cfg-1: 4x1x192x192, 1 step -- also
    pinned on the reference itself at that size: tests/golden/g6l_2d_cfg1_192.npz; cfg-4: 8x1x128x128x64 full chain
    with 3D noise, 5 steps; cfg-5: 4x1x160x160x80 morph-only, rows of 80, 10 steps + the anatomy ladder)."""
    import contextlib
    import io
    import bench
    wl = dict(bench.WORKLOADS[name])
    solver = bench.build_solver(wl, DEV)
    torch.manual_seed(0)
    data = torch.rand(wl["batch"], 1, *wl["dims"], device=DEV)
    model = bench.make_model(len(wl["dims"])).to(DEV)
    kw = bench.solver_kwargs(wl, DEV)
    anat = {k: v for k, v in kw.items() if k.startswith("anatomy") or k.startswith("volume")}
    with contextlib.redirect_stdout(io.StringIO()):
        loss0 = solver.adversarial_training(data=data, model=model, n_iter=0, **anat)
        params0 = [t.param.clone() for t in solver.chain_of_transforms]
        loss1 = solver.adversarial_training(data=data, model=model, lazy_load=True, **kw)
    assert torch.isfinite(loss0) and torch.isfinite(loss1) and float(loss1) > 0
    assert solver.adv_data.shape == data.shape and torch.isfinite(solver.adv_data).all()
    for t, p0 in zip(solver.chain_of_transforms, params0):
        assert torch.isfinite(t.param).all()
        if not wl.get("anatomy"):                  # (the anatomy ladder may end on a fresh re-initialisation)
            assert not torch.equal(t.param, p0)
        if t.get_name() in ("noise", "morph"):     # rescaled to the unit L2 ball per sample at the end
            norms = t.param.reshape(t.param.shape[0], -1).norm(dim=1)
            assert float((norms - 1).abs().max()) < 1e-4
        if t.get_name() == "bias":
            assert float(t.param.max()) <= t.high + 1e-6 and float(t.param.min()) >= t.low - 1e-6
    if wl.get("anatomy"):
        # what the ladder guarantees (adv_compose_solver.py:369-403): either the round trip of the anatomy mask is
        # preserved to the tolerance, or the search gave up after 3 x n_iter steps
        err = float(solver.compute_anatomy_misoverlapping_loss(kw["anatomy_mask_images"]))
        assert err <= kw["volume_preserve_tolerance"] or len(solver.chain_of_transforms) == 2


def _one_step_vs_oracle(sd, N, dims, names, morph_div8=False, seed=11):
    """One whole adversarial_training call (one ascent step + the final consistency pass), same initial parameters on
    both sides.  Tolerances: 1e-4 (scale-relative), widened ONLY by what the oracle itself moves when its own
    deformation fields are jittered by the measured GPU-vs-oracle field difference (interpolation kinks: the
    derivative of the (tri)linear interpolant jumps at grid nodes; tests/golden/g8_kinks.npz holds the same evidence
    from the reference)."""
    from oracle import advchain_oracle as O
    from advchain_amd.augmentor import AdvAffine, AdvBias, AdvMorph, AdvNoise, ComposeAdversarialTransformSolver
    from tests.helpers import make_model, maxdiff, rand, smooth_data
    import bench
    specs = bench.transform_configs(dims, N, names, morph_div8=morph_div8)
    data = smooth_data(N, 1, dims, seed)
    ocls = {"noise": O.OracleNoise, "bias": O.OracleBias, "morph": O.OracleMorph, "affine": O.OracleAffine}
    gcls = {"noise": AdvNoise, "bias": AdvBias, "morph": AdvMorph, "affine": AdvAffine}
    gchain = [gcls[nm](spatial_dims=sd, config_dict=cfg, device=DEV) for nm, cfg in specs]
    init = []
    for i, (nm, cfg) in enumerate(specs):
        o = ocls[nm](sd, cfg)
        o.init_parameters()
        shape = tuple(o.param.shape)
        if nm == "bias":
            p = 0.1 * rand(shape, 200 + i)          # |log field| well inside log(1 +- eps): no clip sub-gradient flips
        elif nm == "affine":
            p = 0.6 * rand(shape, 200 + i)
        else:
            p = O.unit_normalize(rand(shape, 200 + i))
        init.append(p)

    def oracle_run(hook):
        chain = [ocls[nm](sd, cfg) for nm, cfg in specs]
        for o, p in zip(chain, init):
            o.init_parameters()
            o.param = p.clone()
            if o.get_name() == "morph":
                o.field_hook = hook
        solver = O.OracleSolver(chain)
        loss = solver.adversarial_training(data=data, model=make_model(sd), n_iter=1, lazy_load=True, step_sizes=1)
        return solver, chain, float(loss)
    osolver, ochain, oloss = oracle_run(None)
    for g, p in zip(gchain, init):
        g.init_parameters()
        g.set_parameters(p.to(DEV))
    # how far is the GPU's deformation field from the oracle's (normalised units)?
    dq = 0.0
    for o, g in zip(ochain, gchain):
        if g.get_name() == "morph":
            with torch.no_grad():
                om = ocls["morph"](sd, o.config_dict)
                om.init_parameters()
                om.param = init[gchain.index(g)].clone()
                # (the product stores the field un-clamped: the sampler applies clamp(-1, 1) on load)
                dq = max(dq, maxdiff(torch.clamp(g._field(1.0), -1, 1).cpu(), om._field(1)),
                         maxdiff(torch.clamp(g._field(-1.0), -1, 1).cpu(), om._field(-1)))
    assert dq < 2e-5, dq
    gsolver = ComposeAdversarialTransformSolver(chain_of_transforms=gchain)
    gloss = gsolver.adversarial_training(data=data.to(DEV), model=make_model(sd, device=DEV), n_iter=1, lazy_load=True,
                                         step_sizes=1)
    # step-0 distance (before any update): pure forward parity
    d0 = osolver.trace[0]["dist"]
    assert abs(float(gsolver.last_inner_dist) - d0) < 1e-7 + 1e-4 * abs(d0)
    # the oracle's own spread under a field jitter of that amplitude
    spread_p, spread_l = [0.0] * len(specs), 0.0
    if dq > 0:
        for trial in range(2):
            _, jchain, jloss = oracle_run(O.uniform_jitter(dq, seed=trial))
            for i, (o, j) in enumerate(zip(ochain, jchain)):
                spread_p[i] = max(spread_p[i], maxdiff(o.param.detach(), j.param.detach()))
            spread_l = max(spread_l, abs(jloss - oloss))
    for i, (nm_cfg, o, g) in enumerate(zip(specs, ochain, gchain)):
        ref = o.param.detach()
        allowed = max(1e-4 * max(1.0, float(ref.abs().max())), 3.0 * spread_p[i])
        err = float((g.param.detach().cpu() - ref).abs().max())
        assert err < allowed, (nm_cfg[0], "err %.2e allowed %.2e (oracle spread %.2e at field diff %.2e)"
                               % (err, allowed, spread_p[i], dq))
    allowed = max(1e-6 + 1e-4 * abs(oloss), 3.0 * spread_l)
    assert abs(float(gloss) - oloss) < allowed, (float(gloss), oloss, allowed, dq)


@pytest.mark.parametrize("sd", [2, 3])
def test_one_ascent_step_matches_the_oracle_at_realistic_size(sd):
    """At the bench resolution (2D 256 x 256, 3D 64 x 64 x 32; small batches so the CPU oracle finishes in seconds): the
    large-size kernel paths (window scatter, gather form, tiled scatter, marching loss) against the oracle itself, not
    only through properties."""
    if sd == 2:
        _one_step_vs_oracle(2, 4, (256, 256), ["noise", "bias", "morph", "affine"])
    else:
        _one_step_vs_oracle(3, 1, (64, 64, 32), ["bias", "morph", "affine"])


def test_one_ascent_step_matches_the_oracle_at_cfg5_row_length():
    """cfg-5's geometry class: rows of 80 voxels (not a multiple of 64: partial waves in the row kernels), morph-only with
    vector_size = size / 8, at 1x1x40x40x80 so that the CPU oracle finishes in seconds."""
    _one_step_vs_oracle(3, 1, (40, 40, 80), ["morph"], morph_div8=True, seed=13)
