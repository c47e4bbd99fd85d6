"""Host logic of the product on CPU: the transform classes, the solver's control flow, the band tables and the
loss normalisers run against a TEST-ONLY operator backend (tests/cpu_backend.py, built from the oracle's torch
ops) and are checked against the golden vectors.  The HIP kernels themselves are covered by the -m gpu tests."""
import contextlib
import io

import numpy as np
import pytest
import torch

from oracle import advchain_oracle as O
from tests import cpu_backend
from tests.helpers import Fixture, counted_torch_seed, fixture_model, make_model, maxdiff, rand, smooth_data

CPU = torch.device("cpu")


@pytest.fixture
def cpu_ops(monkeypatch):
    cpu_backend.install(monkeypatch)
    yield


def build_chain(spec):
    from advchain_amd.augmentor import AdvAffine, AdvBias, AdvMorph, AdvNoise
    cls = {"noise": AdvNoise, "bias": AdvBias, "morph": AdvMorph, "affine": AdvAffine}
    return [cls[s["name"]](spatial_dims=len(s["config"]["data_size"]) - 2, config_dict=s["config"], device=CPU,
                           **s.get("kwargs", {})) for s in spec]


def test_band_tables_reproduce_the_reference_bias_field(cpu_ops):
    from advchain_amd.augmentor import AdvBias
    fx = Fixture("g2_bias")
    for key, m in fx.json().items():
        t = AdvBias(spatial_dims=m["spatial_dims"], config_dict=m["config"], device=CPU)
        t.init_parameters()
        assert list(t.param.shape) == m["cp_grid"]
        assert t._crop_start.tolist() == m["crop_start"] and t._crop_end.tolist() == m["crop_end"]
        assert t._stride == m["stride"]
        assert max(t._tables.B) <= 5
        p = fx.t(key + "param").requires_grad_(True)
        t.param = p
        if key + "data" in fx:
            data, w = fx.t(key + "data"), fx.t(key + "w")
        else:
            ds = m["config"]["data_size"]
            data = smooth_data(ds[0], ds[1], ds[2:], int(fx.arr(key + "data_seed")))
            w = rand(tuple(data.shape), int(fx.arr(key + "w_seed")))
        o = t.forward(data)
        (o * w).sum().backward()
        if key + "field" in fx:
            assert maxdiff(t.bias_field, fx.t(key + "field")) < 3e-6, key
            assert maxdiff(o, fx.t(key + "out")) < 3e-6, key
        g = fx.t(key + "grad_param")
        assert maxdiff(p.grad, g) < 2e-5 * max(1.0, float(g.abs().max())), key


def test_band_table_structure():
    from advchain_amd import bands
    t = bands.upsample_tables([8, 8, 32], [128, 128, 64], CPU)
    assert t.B == [2, 2, 2] and t.S == [128, 128, 64] and t.g == [8, 8, 32]
    itab = t.itab.numpy()
    start = itab[:128]
    lo, hi = itab[128:136], itab[136:144]
    for k in range(8):
        touched = np.nonzero((start <= k) & (k < start + 2))[0]
        assert lo[k] == touched[0] and hi[k] == touched[-1] + 1
    t2 = bands.upsample_tables([12, 12], [192, 192], CPU)
    assert t2.S == [1, 192, 192] and t2.g == [1, 12, 12] and t2.ndim == 2
    w = np.asarray(t2.ftab.numpy()[1:1 + 192 * 2]).reshape(192, 2)
    assert np.allclose(w.sum(1), 1.0, atol=1e-6)


def test_morph_and_affine_host_classes(cpu_ops):
    from advchain_amd.augmentor import AdvAffine, AdvMorph
    fx = Fixture("g3_morph")
    for key, m in fx.json().items():
        t = AdvMorph(spatial_dims=m["spatial_dims"], config_dict=m["config"], device=CPU)
        t.init_parameters()
        p = fx.t(key + "param").requires_grad_(True)
        t.param = p
        data, w = fx.t(key + "data"), fx.t(key + "w")
        dxy, disp = t.get_deformation_displacement_field(duv=t.epsilon * p)
        assert maxdiff(dxy, fx.t(key + "dxy_fwd")) < 2e-6
        # the reference's own call form (adv_morph.py:342-343): the identity grid as initial deformation, positional
        assert maxdiff(t.DemonsCompose(t.epsilon * p, t.base_grid, smooth=True), dxy) == 0
        assert maxdiff(t.DemonsCompose(duv=t.epsilon * p, init_deformation_dxy=t.base_grid.clone()), dxy) == 0
        # another initial deformation / no final smoothing: the general route (adv_morph.py:474-490 on the chain's positions)
        dims = m["config"]["data_size"][2:]
        other = 0.5 * t.base_grid
        assert maxdiff(t.DemonsCompose(t.epsilon * p, other, smooth=True),
                       O.demons_compose(t.epsilon * p, dims, init=other)) < 2e-6
        assert maxdiff(t.DemonsCompose(t.epsilon * p, t.base_grid, smooth=False),
                       O.demons_compose(t.epsilon * p, dims, smooth=False)) < 2e-6
        with pytest.raises(ValueError):
            t.DemonsCompose(t.epsilon * p, t.base_grid[:, :1])
        t.integration_type = 'euler'          # adv_morph.py:136-141 (2D); the reference's 3D loop raises (adv_morph.py:171)
        if m["spatial_dims"] == 2:
            assert maxdiff(t.DemonsCompose(t.epsilon * p, t.base_grid),
                           O.demons_compose(t.epsilon * p, dims, integration_type='euler')) < 2e-6
        else:
            with pytest.raises(TypeError):
                t.DemonsCompose(t.epsilon * p, t.base_grid)
        t.integration_type = 'ss'
        o = t.forward(data)
        (o * w).sum().backward()
        assert maxdiff(o, fx.t(key + "forward")) < 2e-6
        assert maxdiff(p.grad, fx.t(key + "grad_param_fwd")) < 1e-4 * float(fx.t(key + "grad_param_fwd").abs().max())
        assert maxdiff(t.backward(data), fx.t(key + "backward")) < 2e-6
        assert t.displacement.shape[-1] == m["spatial_dims"]
        assert maxdiff(t.diff, (o - data).detach()) == 0
    fx = Fixture("g4_affine")
    for key, m in fx.json().items():
        t = AdvAffine(spatial_dims=m["spatial_dims"], config_dict=m["config"], device=CPU)
        t.init_parameters()
        p = fx.t(key + "param").requires_grad_(True)
        t.param = p
        data, w = fx.t(key + "data"), fx.t(key + "w")
        o = t.forward(data)
        assert maxdiff(t.affine_matrix, fx.t(key + "theta")) < 1e-6
        assert maxdiff(t.get_inverse_matrix(t.affine_matrix), fx.t(key + "theta_inv")) < 2e-6
        (o * w).sum().backward()
        assert maxdiff(o, fx.t(key + "forward")) < 2e-6
        assert maxdiff(t.backward(data), fx.t(key + "backward")) < 2e-6
        # Q9: a caller-supplied padding mode is ignored by AdvAffine
        assert maxdiff(t.forward(data, padding_mode="border"), o) == 0


def test_loss_normalisers_match_the_reference(cpu_ops):
    from advchain_amd.common.loss import calc_segmentation_consistency
    fx = Fixture("g5_loss")
    for tag in ("2d", "3d"):
        ref, mask = fx.t(tag + "_ref"), fx.t(tag + "_mask")
        for name, types, weights in (("mse", ["mse"], [1.0]), ("contour", ["contour"], [1.0]), ("kl", ["kl"], [1.0]),
                                     ("mix", ["mse", "contour"], [1.0, 0.5])):
            for mtag, mk in (("masked", mask), ("nomask", None)):
                pred = fx.t(tag + "_pred").requires_grad_(True)
                v = calc_segmentation_consistency(pred, ref, types, weights, scales=[0], mask=mk)
                v.backward()
                k = "%s_%s_%s_" % (tag, name, mtag)
                assert abs(float(v) - fx.f(k + "value")) < 1e-7 + 1e-5 * abs(fx.f(k + "value")), k
                g = fx.t(k + "grad")
                assert maxdiff(pred.grad, g) < 1e-5 * float(g.abs().max()) + 1e-10, k


G6_CASES = ["2d_full_n1", "2d_full_n3", "2d_full_n2_norm", "2d_smart_n2", "2d_power_n2", "2d_kl_n1",
            "2d_photometric_n2", "2d_step_n2", "3d_bma_n2", "3d_full_n1", "3d_morph_anat_n2", "2d_n0",
            "2d_bn_drop_train_n2", "3d_bn_drop_eval_n2", "2d_bias_gauss_c2_n2", "2d_isgt_n1"]


def product_dropout_classes():
    from advchain_amd.common.layers import Fixable2DDropout, Fixable3DDropout
    return Fixable2DDropout, Fixable3DDropout


def check_model_state(model, fx):
    """a11: the BatchNorm statistics saw exactly the reference's updates (only the final model.train() pass of
    calc_adv_consistency_loss tracks them) and the Fixable*Dropout seed / lazy_load ended where the reference's did."""
    assert maxdiff(model[1].running_mean.cpu(), fx.t("bn_running_mean")) < 2e-5
    assert maxdiff(model[1].running_var.cpu(), fx.t("bn_running_var")) < 2e-5
    assert int(model[1].num_batches_tracked) == int(fx.arr("bn_num_batches"))
    assert int(model[2].seed) == int(fx.arr("dropout_seed"))
    assert bool(model[2].lazy_load) == bool(fx.arr("dropout_lazy_load"))
    assert bool(model.training) == bool(fx.arr("model_training"))


@pytest.mark.parametrize("case", G6_CASES)
def test_solver_control_flow_reproduces_reference_runs(cpu_ops, case):
    """Product solver + transform classes (oracle-backed ops) vs the reference's own adversarial_training runs."""
    from advchain_amd.augmentor import ComposeAdversarialTransformSolver
    fx = Fixture("g6_" + case)
    meta = fx.json()
    chain = build_chain(meta["chain"])
    for i, t in enumerate(chain):
        t.init_parameters()
        t.set_parameters(fx.t("init_param_%d" % i))
    solver = ComposeAdversarialTransformSolver(chain_of_transforms=chain, **meta["solver"])
    kw = dict(meta["train"])
    if meta["has_anatomy"]:
        kw["anatomy_mask_images"] = fx.t("anatomy")
    model = fixture_model(meta, product_dropout_classes())
    with contextlib.redirect_stdout(io.StringIO()), counted_torch_seed(1000):
        loss = solver.adversarial_training(data=fx.t("data"), model=model, **kw)
    if meta.get("model"):
        check_model_state(model, fx)
    assert abs(float(loss) - fx.f("final_loss")) < 1e-7 + 2e-4 * abs(fx.f("final_loss"))
    assert maxdiff(solver.adv_data, fx.t("adv_data")) < 1e-4
    assert len(solver.chain_of_transforms) == meta["n_transforms"]
    for i, t in enumerate(chain):
        assert maxdiff(t.param, fx.t("final_param_%d" % i)) < 1e-4, (case, i)
        assert not t.param.requires_grad and not t.is_training


def test_nan_guard_skips_the_update(cpu_ops):
    from advchain_amd.augmentor import AdvNoise, ComposeAdversarialTransformSolver
    ds = [2, 1, 8, 8]
    t = AdvNoise(spatial_dims=2, config_dict=dict(epsilon=1.0, xi=1e-6, data_size=ds), device=CPU)
    solver = ComposeAdversarialTransformSolver(chain_of_transforms=[t], divergence_types=["mse"], divergence_weights=[1.0])

    class Bad(torch.nn.Module):
        def forward(self, x):
            return torch.cat([x, x * float("nan")], dim=1)
    t.init_parameters()
    before = t.param.clone()
    data = torch.rand(*ds)
    init = torch.cat([data, data], dim=1)
    solver.optimizing_transform(model=Bad(), data=data, init_output=init, optimize_flags=[True], n_iter=1, step_sizes=[1])
    assert maxdiff(t.param, O_unit(before)) < 1e-6  # no ascent step; only the final rescale (unit L2) happened


def test_device_nan_guard_is_for_the_built_in_updates_only(cpu_ops):
    """a subclass that overrides optimize_parameters() may never hand `_gate` to the gated kernels -- it must get
    the reference's host-side check (adv_compose_solver.py:343-347); and the gate is cleared even when the step raises."""
    from advchain_amd.augmentor import AdvNoise
    from advchain_amd.augmentor.adv_compose_solver import ComposeAdversarialTransformSolver, _native_update
    ds = [2, 1, 8, 8]
    cfg = dict(epsilon=1.0, xi=1e-6, data_size=ds)

    class OwnStep(AdvNoise):
        def optimize_parameters(self, step_size=None):
            self.param = (self.param + 1.0).detach()

    class Plain(AdvNoise):
        pass

    a, b, c = AdvNoise(2, cfg, device=CPU), OwnStep(2, cfg, device=CPU), Plain(2, cfg, device=CPU)
    assert _native_update(a) and _native_update(c) and not _native_update(b)

    class Raises(AdvNoise):
        def optimize_parameters(self, step_size=None):
            assert self._gate is not None
            raise RuntimeError("boom")

    class FakeCuda(torch.Tensor):      # a CPU tensor that claims to live on the GPU: the device-guard branch on this box
        is_cuda = True

    t = Raises(2, cfg, device=CPU)
    t.optimize_parameters = AdvNoise.optimize_parameters.__get__(t)       # instance attribute: the CLASS still overrides
    assert not _native_update(t)
    t2 = AdvNoise(2, cfg, device=CPU)
    solver = ComposeAdversarialTransformSolver(chain_of_transforms=[t2], divergence_types=["mse"], divergence_weights=[1.0])
    solver._global_value = lambda d: d.as_subclass(FakeCuda)
    t2.init_parameters()
    boom = {"n": 0}

    def failing_backward(dist, flags):
        boom["n"] += 1
        assert t2._gate is not None           # the gate is armed while the step runs ...
        raise RuntimeError("backward failed")
    solver._backward_to_transforms = failing_backward
    data = torch.rand(*ds)
    with pytest.raises(RuntimeError, match="backward failed"):
        solver.optimizing_transform(model=make_model(2), data=data, init_output=make_model(2)(data).detach(),
                                    optimize_flags=[True], n_iter=1, step_sizes=[1])
    assert boom["n"] == 1 and t2._gate is None      # ... and cleared although the step raised


def O_unit(x):
    from oracle import advchain_oracle as O
    return O.unit_normalize(x)


def test_anatomy_retry_logic(cpu_ops):
    """adv_compose_solver.py:369-403: +1 step, re-init at 2N, give up at 3N (with the duplicated last transform)."""
    from advchain_amd.augmentor import AdvAffine, ComposeAdversarialTransformSolver
    ds = [1, 1, 16, 16]
    cfg = dict(rot=0.5, scale_x=0.5, scale_y=0.5, shift_x=0.9, shift_y=0.9, data_size=ds)
    t = AdvAffine(spatial_dims=2, config_dict=cfg, device=CPU)
    solver = ComposeAdversarialTransformSolver(chain_of_transforms=[t])
    calls = {"n": 0}
    solver.compute_anatomy_misoverlapping_loss = lambda anatomy_mask_images=None: (calls.__setitem__("n", calls["n"] + 1), torch.tensor(1.0))[1]
    torch.manual_seed(0)
    data = torch.rand(*ds)
    anat = (torch.rand(*ds) > 0.5).float()
    steps = {"n": 0}
    orig = t.optimize_parameters
    t.optimize_parameters = lambda step_size=None: (steps.__setitem__("n", steps["n"] + 1), orig(step_size=step_size))[1]
    with contextlib.redirect_stdout(io.StringIO()):
        solver.adversarial_training(data=data, model=make_model(2), n_iter=1, anatomy_mask_images=anat,
                                    volume_preserve_tolerance=1e-9)
    assert steps["n"] == 3                         # N=1: tries at i=1 (+new init, +N), i=2 (+1), i=3 (>=3N: give up)
    assert len(solver.chain_of_transforms) == 2    # reference quirk: last transform appended again on give-up


def test_displacement_hint_bits_and_halo_policies():
    """Host-side policy helpers of advchain_amd.ops (no kernel call): the forward displacement hint travels in bits 8..15
    of an int argument; exact bounds are only promised below the measured displacement; the 3D policy stops at 4 voxels
    (window scatter beyond), the 2D one at 16 px."""
    from advchain_amd import ops
    assert ops._hint_bits(None) == 0 and ops._hint_bits(float("nan")) == 0 and ops._hint_bits(-1.0) == 0
    assert ops._hint_bits(0.3) == 1 << 8 and ops._hint_bits(1.0) == 2 << 8 and ops._hint_bits(6.2) == 7 << 8
    assert ops._hint_bits(1e9) == 255 << 8
    assert [ops.squaring_halo(x, 3) for x in (0.5, 0.9995, 1.5, 2.5, 3.5, 3.9995, 7.0)] == [-1, -2, -2, -3, -4, 8, 8]
    # (2D squarings: the whole-row scatter up to an exact 32 px; image warps stop at 16)
    assert [ops.squaring_halo(x, 2) for x in (0.5, 1.5, 2.5, 3.0, 5.0, 7.9, 11.0, 15.0, 15.9995, 23.0, 31.0, 31.9995, 40.0)] == \
        [-1, -2, -3, -4, -6, -8, -12, -16, -24, -24, -32, 16, 16]
    assert [ops.warp_halo([None, x, 0, 0], 2) for x in (15.0, 15.9995, 31.0)] == [-16, 16, 16]
    assert ops.squaring_halo(float("nan"), 3) == 0
    assert ops.warp_halo([None, 0.4, 0, 0], 3) == -1 and ops.warp_halo([None, 5.0, 0, 0], 3) == 8
    assert ops.warp_halo([None, 0.4, 0, 0], 2) == -2 and ops.warp_halo([None, 20.0, 0, 0], 2) == 16
