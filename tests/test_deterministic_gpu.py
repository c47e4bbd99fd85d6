"""Deterministic mode (round 6; include/advchain_hip.h: advchain_set_deterministic, ComposeAdversarialTransformSolver(deterministic=...)).

The one backward formulation whose bits depend on arrival order is the source-tiled window scatter (scatter_window.hip: 2D
image warps above 16 px, squarings above 32 px; 3D above 4 voxels), which flushes its LDS windows with float atomics.  In
deterministic mode the tiles add 64-bit fixed point into an int64 image of grad_in instead and a last pass converts it.

What must hold:
  * the deterministic twin computes the same gradients as autograd through F.grid_sample (the reference's
    grid_sampler_*_backward via adv_morph.py:546-557) to the tolerance of the float-atomic form;
  * two runs of the same call are equal BIT FOR BIT -- operators, and whole solver calls at the headline shape (cfg-2, five
    ascent steps: its image warps reach 30-70 px) and on a cfg-5-shaped 3D morph chain (fields of 5-10 voxels);
  * a non-finite gradient is not dropped; a batch entry's result does not depend on what else is in the batch.
"""
import contextlib
import io

import pytest
import torch
import torch.nn.functional as F

from tests.helpers import maxdiff, rand
from tests.test_ops_gpu import _smooth_field

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")


@pytest.fixture
def det():
    from advchain_amd import ops
    ops.set_deterministic(True)
    try:
        yield ops
    finally:
        ops.set_deterministic(False)


def test_workspace_size_follows_the_mode(det):
    ops = det
    from advchain_amd import _lib
    lib = _lib.load()
    dims = (12, 20, 16)
    big = lib.advchain_scatter_workspace(3, 3, _lib.dims_array(dims))
    ops.set_deterministic(False)
    small = lib.advchain_scatter_workspace(3, 3, _lib.dims_array(dims))
    V = 12 * 20 * 16
    assert small == 4 + 2 * 3 * V and big == small + 8 * 3 * V + 4, (small, big)
    assert not ops.is_deterministic()


@pytest.mark.parametrize("dims", [(9, 18, 64), (13, 21, 80), (20, 30, 44)])
@pytest.mark.parametrize("amp_vox", [4.5, 9.0])
def test_window_scatter_3d_deterministic(det, dims, amp_vox):
    """3D, displacement hint beyond the owner-computes march: self-composition and image warps (C = 1, 2, 4; both paddings;
    clamped grid; with and without grad_grid) against autograd, twice, bit for bit.  9 voxels on these small volumes stretches
    the windows past the LDS budget: the deposits outside a capped window take the fixed-point route as well."""
    from oracle import advchain_oracle as O
    ops = det
    d, halo = 3, 8
    phi = _smooth_field(dims, amp_vox, 61)
    w = rand((2, d) + dims, 62)
    p = phi.clone().requires_grad_(True)
    (O.compose_fields(p, p) * w).sum().backward()
    pd, wd = phi.to(DEV), w.to(DEV)
    ws = ops._scatter_workspace(2, dims, DEV)
    g1 = ops.raw_compose_self_bwd(wd, pd, ws, chain=False, halo=halo)
    assert maxdiff(g1.cpu(), p.grad) < 5e-5 * max(1.0, float(p.grad.abs().max()))
    for _ in range(3):
        assert torch.equal(g1, ops.raw_compose_self_bwd(wd, pd, ws, chain=False, halo=halo))
    g2 = ops.raw_compose_self_bwd(g1, pd, ws, chain=True, halo=0)          # owner-computes tiles after a window launch
    p2 = phi.clone().requires_grad_(True)
    (O.compose_fields(p2, p2) * p.grad).sum().backward()
    assert maxdiff(g2.cpu(), p2.grad) < 2e-4 * max(1.0, float(p2.grad.abs().max()))
    for C in (1, 2, 4):
        for pad, clamp in (("zeros", True), ("zeros", False), ("border", False)):
            grid = (phi * 1.02).contiguous()
            inp, wv = rand((2, C) + dims, 63 + C), rand((2, C) + dims, 73 + C)
            a, g = inp.clone().requires_grad_(True), grid.clone().requires_grad_(True)
            gp = torch.clamp(g, -1, 1) if clamp else g
            (F.grid_sample(a, gp.permute(0, 2, 3, 4, 1), padding_mode=pad, align_corners=True) * wv).sum().backward()
            args = (wv.to(DEV), inp.to(DEV), grid.to(DEV), 0, ops.pad_code(pad), clamp)
            gin, ggrid = ops.raw_grid_sample_bwd(*args, True, True, halo)
            assert maxdiff(gin.cpu(), a.grad) < 5e-5 * max(1.0, float(a.grad.abs().max())), (C, pad, clamp)
            assert maxdiff(ggrid.cpu(), g.grad) < 5e-5 * max(1.0, float(g.grad.abs().max())), (C, pad, clamp)
            again, gg2 = ops.raw_grid_sample_bwd(*args, True, True, halo)
            assert torch.equal(gin, again) and torch.equal(ggrid, gg2), (C, pad, clamp)
            only, none = ops.raw_grid_sample_bwd(*args, True, False, halo)
            assert none is None and torch.equal(only, gin), (C, pad, clamp)


@pytest.mark.parametrize("dims", [(64, 96), (100, 72), (256, 256)])
@pytest.mark.parametrize("amp_px", [20.0, 45.0])
def test_window_scatter_2d_deterministic(det, dims, amp_px):
    """2D: squarings and image warps beyond the whole-row scatter's bounds (hint 16: the window scatter)."""
    from oracle import advchain_oracle as O
    ops = det
    d, halo = 2, 16
    phi = _smooth_field(dims, amp_px, 41)
    w = rand((2, d) + dims, 42)
    p = phi.clone().requires_grad_(True)
    (O.compose_fields(p, p) * w).sum().backward()
    pd, wd = phi.to(DEV), w.to(DEV)
    ws = ops._scatter_workspace(2, dims, DEV)
    g1 = ops.raw_compose_self_bwd(wd, pd, ws, chain=False, halo=halo)
    assert maxdiff(g1.cpu(), p.grad) < 5e-5 * max(1.0, float(p.grad.abs().max()))
    for _ in range(3):
        assert torch.equal(g1, ops.raw_compose_self_bwd(wd, pd, ws, chain=False, halo=halo))
    for C in (1, 2, 4):
        for pad, clamp in (("zeros", True), ("border", False)):
            grid = (phi * 1.02).contiguous()
            inp, wv = rand((2, C) + dims, 43 + C), rand((2, C) + dims, 53 + C)
            a, g = inp.clone().requires_grad_(True), grid.clone().requires_grad_(True)
            gp = torch.clamp(g, -1, 1) if clamp else g
            (F.grid_sample(a, gp.permute(0, 2, 3, 1), padding_mode=pad, align_corners=True) * wv).sum().backward()
            args = (wv.to(DEV), inp.to(DEV), grid.to(DEV), 0, ops.pad_code(pad), clamp)
            gin, ggrid = ops.raw_grid_sample_bwd(*args, True, True, halo)
            assert maxdiff(gin.cpu(), a.grad) < 5e-5 * max(1.0, float(a.grad.abs().max())), (C, pad, clamp)
            assert maxdiff(ggrid.cpu(), g.grad) < 5e-5 * max(1.0, float(g.grad.abs().max())), (C, pad, clamp)
            again, gg2 = ops.raw_grid_sample_bwd(*args, True, True, halo)
            assert torch.equal(gin, again) and torch.equal(ggrid, gg2), (C, pad, clamp)


def test_deterministic_scatter_keeps_batch_entries_apart_and_surfaces_non_finite_gradients(det):
    """The fixed-point scale is a batch ENTRY's max |grad_out|: entry 0 gives the same bits whether entry 1 is there or not, or
    holds gradients a million times larger; a NaN / inf in entry 1 turns entry 1's outputs into NaN and leaves entry 0 alone
    (the owner-computes scatters' rule, LESSONS 36)."""
    ops = det
    dims, halo = (13, 21, 80), 8
    phi = _smooth_field(dims, 6.0, 71).to(DEV)
    inp, wv = rand((2, 1) + dims, 72).to(DEV), rand((2, 1) + dims, 73).to(DEV)
    both, _ = ops.raw_grid_sample_bwd(wv, inp, phi, 0, 0, True, True, True, halo)
    one, _ = ops.raw_grid_sample_bwd(wv[:1].contiguous(), inp[:1].contiguous(), phi[:1].contiguous(), 0, 0, True, True, True, halo)
    assert torch.equal(both[:1], one)
    big = wv.clone()
    big[1] *= 1e6
    scaled, _ = ops.raw_grid_sample_bwd(big, inp, phi, 0, 0, True, True, True, halo)
    assert torch.equal(scaled[:1], one)
    bad = wv.clone()
    bad[1, 0, 5, 7, 9] = float("nan")
    got, _ = ops.raw_grid_sample_bwd(bad, inp, phi, 0, 0, True, True, True, halo)
    assert torch.equal(got[:1], one) and bool(torch.isnan(got[1]).all())
    bad[1, 0, 5, 7, 9] = float("inf")
    got, _ = ops.raw_grid_sample_bwd(bad, inp, phi, 0, 0, True, True, True, halo)
    assert torch.equal(got[:1], one) and bool(torch.isnan(got[1]).any())


def _solver_run(wl_name, batch, deterministic, seed, n_iter=None, hip_graph=False):
    import bench
    wl = dict(bench.WORKLOADS[wl_name], batch=batch)
    if n_iter is not None:
        wl["n_iter"] = n_iter
    solver = bench.build_solver(wl, DEV, None, hip_graph=hip_graph)
    solver.deterministic = deterministic
    torch.manual_seed(seed)
    data = torch.rand(batch, 1, *wl["dims"], device=DEV)
    model = bench.make_model(len(wl["dims"])).to(DEV)
    kw = bench.solver_kwargs(wl, DEV)
    outs = []
    for rep in range(2):
        torch.manual_seed(seed + 1)
        with contextlib.redirect_stdout(io.StringIO()):
            loss = solver.adversarial_training(data=data, model=model, **kw)
        outs.append([solver.adv_data.detach().clone(), solver.warped_back_adv_output.detach().clone()]
                    + [t.param.detach().clone() for t in solver.chain_of_transforms] + [loss.detach().clone()])
    return solver, outs


def _window_scatter_was_taken(sd):
    """Did the run's backward reach the regime of the window scatter?  (2D: an image warp beyond 16 px or a squaring beyond
    32 px; 3D: beyond 4 voxels -- from the displacement read-backs the run recorded.)"""
    from advchain_amd import ops
    lim_sq, lim_warp = (32.0, 16.0) if sd == 2 else (4.0, 4.0)
    sq = max((max(h[:-1]) for h in ops._CHAIN_HINTS.values() if len(h) > 1), default=0.0)
    warp = max(list(ops._WARP_HINTS.values()) + [h[-1] for h in ops._CHAIN_HINTS.values()], default=0.0)
    return sq >= lim_sq or warp >= lim_warp, (sq, warp)


def test_headline_workload_is_bit_reproducible_in_deterministic_mode():
    """cfg-2 (32 x 1 x 256 x 256, full chain, five ascent steps): from the second ascent step on its image-warp backward is the
    window scatter.  Two calls from the same initial parameters: parameters, adversarial data, warped-back prediction equal bit
    for bit -- and against the default mode to the tolerance between two summation orders of the ascent (chaotic for five
    free-running steps: only the first-step-dominated quantities are held to it)."""
    from advchain_amd import ops
    ops._CHAIN_HINTS.clear(); ops._WARP_HINTS.clear()
    solver, (a, b) = _solver_run("cfg2", 32, True, 5)
    taken, seen = _window_scatter_was_taken(2)
    assert taken, seen
    assert ops.is_deterministic()
    for x, y in zip(a[:-1], b[:-1]):
        assert torch.equal(x, y)
    assert abs(float(a[-1]) - float(b[-1])) <= 1e-6 * abs(float(b[-1]))       # (the loss VALUE: partial sums in arrival order)
    assert all(bool(torch.isfinite(x).all()) for x in a)


def test_cfg5_shaped_morph_chain_is_bit_reproducible_in_deterministic_mode():
    """cfg-5's chain (160 x 160 x 80, morph only, vector_size = dims // 8, anatomy regulariser, ten ascent steps; one volume):
    its squarings and image warps leave the 4-voxel march for the window scatter in every step."""
    from advchain_amd import ops
    ops._CHAIN_HINTS.clear(); ops._WARP_HINTS.clear()
    solver, (a, b) = _solver_run("cfg5", 1, True, 7)
    taken, seen = _window_scatter_was_taken(3)
    assert taken, seen
    for x, y in zip(a[:-1], b[:-1]):
        assert torch.equal(x, y)
    assert all(bool(torch.isfinite(x).all()) for x in a)


def test_deterministic_mode_follows_torch_and_is_part_of_the_graph_key():
    """deterministic=None follows torch.are_deterministic_algorithms_enabled(); a replayed ascent loop is captured per mode."""
    from advchain_amd import ops
    solver, _ = _solver_run("cfg1", 4, None, 3, n_iter=1)
    assert not ops.is_deterministic()
    torch.use_deterministic_algorithms(True, warn_only=True)
    try:
        solver._apply_deterministic(torch.zeros(1, device=DEV))
        assert ops.is_deterministic() and solver._deterministic_now
    finally:
        torch.use_deterministic_algorithms(False)
    solver._apply_deterministic(torch.zeros(1, device=DEV))
    assert not ops.is_deterministic()
    # the replayed loop: deterministic or not is baked into the capture; the key keeps the two apart
    import bench
    wl = dict(bench.WORKLOADS["cfg1"], batch=4)
    g = bench.build_solver(wl, DEV, None, hip_graph=True)
    data = torch.rand(4, 1, *wl["dims"], device=DEV)
    model = bench.make_model(2).to(DEV)
    for mode in (False, True, False):
        g.deterministic = mode
        for _ in range(5):
            with contextlib.redirect_stdout(io.StringIO()):
                g.adversarial_training(data=data, model=model, n_iter=1, step_sizes=1, power_iteration=False)
    assert len(g._graphs) == 2 and g.graph_stats["captures"] == 2, (len(g._graphs), g.graph_stats)
    ops.set_deterministic(False)
