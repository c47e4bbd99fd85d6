"""The validity mask riding through the warps of the data (advchain_grid_sample_fwd_ride / advchain_affine_warp_fwd_ride):
the reference warps the all-ones mask with four calls of its own (adv_compose_solver.py:262-268, 321-325); here it is one more
channel of the launches that warp the data and the prediction.  Every value must be what the separate calls return, bit for
bit -- at the operator level and for whole solver calls (loss, parameters, adversarial data)."""
import pytest
import torch

from tests.helpers import rand, smooth_data

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _grid(N, dims, amp, seed):
    from oracle import advchain_oracle as O
    d = len(dims)
    return (O.identity_grid(N, dims) + amp * rand((N, d) + tuple(dims), seed)).contiguous().to(DEV)


@pytest.mark.parametrize("dims", [(64, 96), (37, 50), (256, 256), (16, 24, 32), (9, 10, 11)])
@pytest.mark.parametrize("interp", ["bilinear", "nearest"])
@pytest.mark.parametrize("padding", ["zeros", "border", "reflection"])
@pytest.mark.parametrize("C", [1, 4, 5])
def test_grid_sample_rider_equals_two_calls(dims, interp, padding, C):
    from advchain_amd import ops
    N = 3
    inp = rand((N, C) + tuple(dims), 1).to(DEV)
    ride = rand((N, 1) + tuple(dims), 2, 0.0, 1.0).to(DEV)
    ride[:, :, :2] = 0.0                                  # (exact zeros for the != 0 flag)
    grid = _grid(N, dims, 0.4, 3)                         # leaves the volume in places
    for clamp in (False, True):
        ref = ops.grid_sample(inp, grid, interp, padding, clamp)
        rref = ops.grid_sample(ride, grid, interp, padding, clamp)
        out, rout = ops.grid_sample(inp, grid, interp, padding, clamp, ride=ride)
        assert torch.equal(out, ref) and torch.equal(rout, rref)
        out, rout = ops.grid_sample(inp, grid, interp, padding, clamp, ride=ride, ride_nonzero=True)
        assert torch.equal(out, ref) and torch.equal(rout, (rref != 0).float())


@pytest.mark.parametrize("dims", [(64, 96), (37, 50), (256, 256), (16, 24, 32), (9, 10, 11), (128, 128, 64)])
@pytest.mark.parametrize("interp", ["bilinear", "nearest"])
@pytest.mark.parametrize("C", [1, 4])
def test_affine_rider_equals_two_calls(dims, interp, C):
    from advchain_amd import ops
    N, d = 3, len(dims)
    inp = rand((N, C) + tuple(dims), 4).to(DEV)
    ride = torch.ones((N, 1) + tuple(dims), device=DEV)
    theta = torch.eye(d, d + 1).repeat(N, 1, 1) + 0.15 * rand((N, d, d + 1), 5)
    theta[2] *= 3.0                                       # strong minification: the tile's box does not fit -> direct gathers
    theta = theta.to(DEV)
    for padding in ("zeros", "border"):
        ref = ops.affine_warp(inp, theta, interp, padding)
        rref = ops.affine_warp(ride, theta, interp, padding)
        out, rout = ops.affine_warp(inp, theta, interp, padding, ride=ride)
        assert torch.equal(out, ref) and torch.equal(rout, rref)
        out, rout = ops.affine_warp(inp, theta, interp, padding, ride=ride, ride_nonzero=True)
        assert torch.equal(out, ref) and torch.equal(rout, (rref != 0).float())


def test_rider_carries_no_gradient_and_the_data_gradient_is_unchanged():
    from advchain_amd import ops
    N, dims = 2, (48, 64)
    grid = _grid(N, dims, 0.02, 7)
    ride = torch.ones((N, 1) + dims, device=DEV)
    gs = []
    for with_ride in (False, True):
        inp = rand((N, 4) + dims, 6).to(DEV).requires_grad_(True)
        g = grid.clone().requires_grad_(True)
        th = (torch.eye(2, 3).repeat(N, 1, 1) + 0.05 * rand((N, 2, 3), 8)).to(DEV).requires_grad_(True)
        if with_ride:
            a, r = ops.grid_sample(inp, g, "bilinear", "zeros", True, ride=ride)
            b, r2 = ops.affine_warp(a, th, "bilinear", "zeros", ride=r, ride_nonzero=True)
            assert not r.requires_grad and not r2.requires_grad
        else:
            b = ops.affine_warp(ops.grid_sample(inp, g, "bilinear", "zeros", True), th, "bilinear", "zeros")
        (b * b).sum().backward()
        gs.append((inp.grad.clone(), g.grad.clone(), th.grad.clone()))
    for x, y in zip(*gs):
        assert torch.equal(x, y)


@pytest.mark.parametrize("case", ["2d_full", "2d_morph_border", "3d_bma", "2d_affine_nearest_back"])
def test_solver_results_do_not_depend_on_the_rider(case):
    """adversarial_training with ops.RIDE_MASK on / off: loss, parameters, adversarial data, warped-back prediction and the
    transforms' `diff` bookkeeping, bit for bit."""
    import bench
    from advchain_amd import ops
    from advchain_amd.augmentor import AdvAffine, AdvBias, AdvMorph, AdvNoise, ComposeAdversarialTransformSolver
    from tests.helpers import make_model
    dims = (32, 32, 16) if case.startswith("3d") else (64, 64)
    names = {"2d_full": ["noise", "bias", "morph", "affine"], "2d_morph_border": ["morph"], "3d_bma": ["bias", "morph", "affine"],
             "2d_affine_nearest_back": ["affine"]}[case]
    N, sd = 2, len(dims)
    cls = {"noise": AdvNoise, "bias": AdvBias, "morph": AdvMorph, "affine": AdvAffine}
    data = smooth_data(N, 1, dims, 5).to(DEV)
    model = make_model(sd, device=DEV)
    outs = {}
    for rider in (False, True):
        ops.RIDE_MASK = rider
        ops._CHAIN_HINTS.clear()
        try:
            torch.manual_seed(3)
            chain = []
            for nm, cfg in bench.transform_configs(dims, N, names):
                kw = {}
                if case == "2d_morph_border":
                    kw["image_padding_mode"] = "border"
                if case == "2d_affine_nearest_back":
                    cfg = dict(cfg, backward_interp="nearest")
                chain.append(cls[nm](spatial_dims=sd, config_dict=cfg, device=torch.device(DEV), **kw))
            solver = ComposeAdversarialTransformSolver(chain_of_transforms=chain, divergence_types=["mse", "contour"],
                                                       divergence_weights=[1.0, 0.5])
            assert solver._ride_ok(chain) == rider
            loss = solver.adversarial_training(data=data, model=model, n_iter=2, lazy_load=False, step_sizes=1,
                                               power_iteration=False)
            outs[rider] = ([loss.detach().clone(), solver.adv_data.clone(), solver.warped_back_adv_output.detach().clone()]
                           + [t.param.detach().clone() for t in chain] + [d.clone() for d in solver.diffs])
        finally:
            ops.RIDE_MASK = True
    for i, (x, y) in enumerate(zip(outs[False], outs[True])):
        assert torch.equal(x, y), (case, i, float((x - y).abs().max()))


def test_special_padding_or_a_subclass_keeps_the_separate_mask_warps():
    """'lowest' / numeric padding, bicubic interpolation and subclasses of the built-in transforms do not ride."""
    import bench
    from advchain_amd.augmentor import AdvAffine, AdvMorph, ComposeAdversarialTransformSolver
    dims, N = (32, 32), 2
    cfgs = dict(bench.transform_configs(dims, N, ["morph", "affine"]))

    class MyMorph(AdvMorph):
        pass
    mk = lambda chain: ComposeAdversarialTransformSolver(chain_of_transforms=chain, divergence_types=["mse", "contour"],
                                                         divergence_weights=[1.0, 0.5])
    dev = torch.device(DEV)
    ok = [AdvMorph(spatial_dims=2, config_dict=cfgs["morph"], device=dev), AdvAffine(spatial_dims=2, config_dict=cfgs["affine"], device=dev)]
    assert mk(ok)._ride_ok(ok)
    cubic = AdvAffine(spatial_dims=2, config_dict=cfgs["affine"], device=dev)
    cubic.backward_interp = "bicubic"
    for bad in ([AdvMorph(spatial_dims=2, config_dict=cfgs["morph"], device=dev, image_padding_mode="lowest")],
                [AdvMorph(spatial_dims=2, config_dict=cfgs["morph"], device=dev, image_padding_mode=0.5)],
                [cubic], [MyMorph(spatial_dims=2, config_dict=cfgs["morph"], device=dev)]):
        assert not mk(bad)._ride_ok(bad)
    s = mk(ok)
    s.divergence_types = ["kl"]
    assert not s._ride_ok(ok)
