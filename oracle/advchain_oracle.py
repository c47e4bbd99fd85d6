"""CPU oracle for the AdvChain adversarial-augmentation inner loop.

TEST INFRASTRUCTURE -- NOT PART OF THE PRODUCT.  Only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may import this module.  The product package
(``advchain_amd``) never imports it and has no CPU fallback.

What it is: a from-scratch, functional restatement (stock fp32 torch-CPU ops) of the
reference's algorithm for the path named in BASELINE.json ``north_star``.  Every function
cites the reference file:line (paths relative to the upstream repo root) that it follows,
including the reference's quirks.

Parity pin: the reference ships no tests or golden vectors for this path, so
the oracle is pinned against OUTPUTS OF THE REFERENCE ITSELF, imported in the build
container by ``oracle/make_golden.py`` and committed as ``tests/golden/*.npz``
(``tests/test_oracle_golden.py`` checks every fixture).  The arithmetic lives in un-vendored
PyTorch ATen ops (``torch>=1.6.0`` per ``requirements.txt:4``; exercised here with
torch 2.10.0): ``grid_sampler_{2,3}d``, ``affine_grid_generator``, ``upsample_{bi,tri}linear``,
``conv_transpose{2,3}d``, ``linalg_inv`` -- the oracle calls the very same CPU ops.
"""
import math
import contextlib

import numpy as np
import torch
import torch.nn.functional as F

CPU = torch.device("cpu")


# --------------------------------------------------------------------------------------
# small shared pieces
# --------------------------------------------------------------------------------------
def unit_normalize(d, p_type="l2"):
    """Per-sample normalisation.  advchain/augmentor/adv_transformation_base.py:129-156."""
    shape = d.shape
    flat = d.reshape(shape[0], -1)
    if p_type == "l1":
        out = flat / flat.norm(p=1, dim=1, keepdim=True)
    elif p_type == "infinity":
        out = flat / (1e-20 + flat.max(dim=1, keepdim=True)[0])
    else:
        out = flat / (flat.norm(dim=1, keepdim=True) + 1e-20)
    return out.reshape(shape)


def identity_grid(n, dims, device=CPU):
    """Channels-first identity sampling grid (N,d,*dims); channel order (x,y[,z]) <-> (last..first)
    spatial dim.  advchain/augmentor/adv_morph.py:14-55."""
    axes = [torch.linspace(-1, 1, s, device=device) for s in dims]
    mesh = torch.meshgrid(axes, indexing="ij")
    chans = [m.unsqueeze(0).unsqueeze(0).repeat(n, 1, *([1] * len(dims))) for m in reversed(mesh)]
    return torch.cat(chans, dim=1)


def _to_sampler_layout(field):
    """(N,d,...) -> (N,...,d) strided view, as adv_morph.py:187,200,535-537."""
    d = field.dim() - 2
    return field.permute(0, *range(2, 2 + d), 1)


# --------------------------------------------------------------------------------------
# AdvMorph numerics
# --------------------------------------------------------------------------------------
def gaussian_window(spatial_dims, sigma=1, kernel_size=5):
    """Normalised dense Gaussian window.  adv_morph.py:391-428 (Q3: k forced to 2*int(4s+.5)+1;
    2D uses '<', 3D '<=')."""
    full = 2 * int(4 * sigma + 0.5) + 1
    if spatial_dims == 2:
        if kernel_size < full:
            kernel_size = full
    else:
        if kernel_size <= full:
            kernel_size = full
    coords = torch.arange(kernel_size).float()
    mean = (kernel_size - 1) / 2.0
    mesh = torch.meshgrid([coords] * spatial_dims, indexing="ij")
    r2 = sum((m - mean) ** 2.0 for m in mesh)
    w = torch.exp(-r2 / (2 * float(sigma) ** 2.0))
    return w / w.sum()


def gaussian_smooth(x, sigma=1, kernel_size=5):
    """Depthwise dense Gaussian conv with zero padding.  adv_morph.py:377-389,430-452."""
    d = x.dim() - 2
    ch = x.shape[1]
    w = gaussian_window(d, sigma, kernel_size).to(x.device)
    k = w.shape[0]
    weight = w.reshape(1, 1, *w.shape).repeat(ch, 1, *([1] * d))
    conv = F.conv2d if d == 2 else F.conv3d
    return conv(x, weight, bias=None, stride=1, padding=k // 2, groups=ch)


def compose_fields(flow1, flow2):
    """flow1 sampled at flow2, linear, border.  adv_morph.py:179-202."""
    return F.grid_sample(flow1, _to_sampler_layout(flow2), padding_mode="border", align_corners=True)


def field_exponentiation(u, nb_steps=8, integration_type="ss"):
    """Scaling and squaring ('ss'), or nb_steps Euler compositions with phi_0 for any other type (2D only in the
    reference: adv_morph.py:136-141; its 3D loop, :171, calls range() on a float).  adv_morph.py:116-177.

    Q1: the start grid is aliased with phi_0 = id + u/2^n (in-place add at adv_morph.py:111), so
    the function returns phi_n - phi_0, not phi_n - id.  Q2: 3D only -- n grows while the
    whole-batch Frobenius norm of u/2^n exceeds 0.5 (adv_morph.py:160-162)."""
    d = u.dim() - 2
    step = u / (2.0 ** nb_steps)
    if d == 3:
        while torch.norm(step) > 0.5:
            nb_steps += 1
            step = u / (2.0 ** nb_steps)
    phi0 = identity_grid(u.shape[0], u.shape[2:], u.device) + step
    phi = phi0
    if integration_type == "ss":
        for _ in range(nb_steps):
            phi = compose_fields(phi, phi)
    else:
        if d == 3:
            raise TypeError("'float' object cannot be interpreted as an integer")    # adv_morph.py:171
        for _ in range(nb_steps):
            phi = compose_fields(phi0, phi)
    return phi - phi0, nb_steps


def demons_compose(duv, dims, final_clamp=True, num_steps=8, smooth_iter=1, sigma=1, smooth=True, init=None,
                   integration_type="ss", gaussian_ks=5):
    """Low-res velocity (N,d,v...) -> clamped sampling grid (N,d,*dims).  adv_morph.py:454-491 (Q4).
    ``final_clamp=False`` (test aid) stops before the last torch.clamp of adv_morph.py:490.  num_steps / smooth_iter /
    sigma / gaussian_ks: the attributes of adv_morph.py:236-240; ``smooth`` and ``init`` (None = the identity grid, what every call of
    the reference passes): the arguments of adv_morph.py:454."""
    d = len(dims)
    base = identity_grid(duv.shape[0], dims, duv.device)
    for _ in range(smooth_iter):                                   # adv_morph.py:386-387
        duv = gaussian_smooth(duv, sigma=sigma, kernel_size=gaussian_ks)
    duv = F.interpolate(duv, size=tuple(dims), mode="bilinear" if d == 2 else "trilinear",
                        align_corners=False)
    offsets, _ = field_exponentiation(duv, num_steps, integration_type)
    composed = compose_fields(base if init is None else init, offsets + base)
    if smooth:
        composed = gaussian_smooth(composed - base, sigma=sigma, kernel_size=gaussian_ks) + base
    return torch.clamp(composed, -1, 1) if final_clamp else composed


def _warp_with_padding(data, grid_nhwc, interp, padding_mode):
    """Shared 'lowest'/numeric padding handling.  adv_morph.py:542-557, adv_affine.py:299-313."""
    if padding_mode == "lowest":
        pad = data.reshape(data.size(0), -1).min(dim=1, keepdim=True).values.detach().clone()
        return F.grid_sample(data - pad, grid_nhwc, mode=interp, align_corners=True,
                             padding_mode="zeros") + pad
    if isinstance(padding_mode, (float, int)):
        return F.grid_sample(data - padding_mode, grid_nhwc, mode=interp, align_corners=True,
                             padding_mode="zeros") + padding_mode
    return F.grid_sample(data, grid_nhwc, mode=interp, align_corners=True, padding_mode=padding_mode)


# --------------------------------------------------------------------------------------
# AdvBias numerics
# --------------------------------------------------------------------------------------
def bspline_window(spacing, order):
    """Dense B-spline window by iterated box filtering.  adv_bias.py:12-49 (Q5: the 2D variant
    pads by i*s per round and ends up zero-padded to 10s+3; the 3D one pads s-1 -> 4s-3)."""
    d = len(spacing)
    ones = torch.ones(1, 1, *spacing)
    k = ones
    vol = float(np.prod(spacing))
    if d == 2:
        for i in range(1, order + 1):
            k = F.conv2d(k, ones, padding=[i * s for s in spacing]) / vol
    else:
        for _ in range(1, order + 1):
            k = F.conv3d(k, ones, padding=[s - 1 for s in spacing]) / vol
    return k[0, 0].to(torch.float32)


class BiasGeometry(object):
    """Control-point lattice / crop arithmetic.  adv_bias.py:84-102,202-277 (Q6)."""

    def __init__(self, data_size, control_point_spacing, downscale, order):
        self.image_size = np.array(data_size[2:])
        self.downscale = downscale
        assert downscale <= min(data_size[2:])
        self.spacing = [s // downscale for s in control_point_spacing]
        stride = np.array(self.spacing)
        low = self.image_size / (1.0 * downscale)
        cp = np.ceil(np.divide(low, stride)).astype(int)
        inner = np.multiply(stride, cp) - (stride - 1)
        cp = cp + 2
        diff = inner - low
        fl = np.floor(np.abs(diff) / 2) * np.sign(diff)
        self.crop_start = (fl + np.remainder(diff, 2) * np.sign(diff)).astype(int)
        self.crop_end = fl.astype(int)
        self.cp_grid = [data_size[0], 1] + cp.tolist()
        self.stride = stride.astype(int).tolist()
        self.window = bspline_window(self.spacing, order)
        self.padding = ((np.array(self.window.shape) - 1) / 2).astype(int).tolist()


def bias_field(cpoint, geom, use_log=True):
    """Control points -> full-res multiplicative field (before clipping).  adv_bias.py:279-335 (Q7)."""
    d = len(geom.stride)
    w = geom.window.to(cpoint.device).unsqueeze(0).unsqueeze(0)
    st, cs, ce = geom.stride, geom.crop_start, geom.crop_end
    if d == 2:
        f = F.conv_transpose2d(cpoint, w, padding=geom.padding, stride=st, groups=1)
        f = f[:, :, st[0] + cs[0]:-st[0] - ce[0], st[1] + cs[1]:-st[1] - ce[1]]
        sh, sw = geom.image_size[0] / f.size(2), geom.image_size[1] / f.size(3)
        if sh > 1 or sw > 1:
            f = F.interpolate(f, size=(int(geom.image_size[0]), int(geom.image_size[1])),
                              mode="bilinear", align_corners=False)
    else:
        f = F.conv_transpose3d(cpoint, w, padding=geom.padding, stride=st, groups=1)
        f = f[:, :, st[0] + cs[0]:-st[0] - ce[0], st[1] + cs[1]:-st[1] - ce[1],
              st[2] + cs[2]:-st[2] - ce[2]]
        sf = tuple(float(geom.image_size[i] / f.size(2 + i)) for i in range(3))
        if max(sf) > 1:
            f = F.interpolate(f, scale_factor=sf, mode="trilinear", align_corners=False)
    return torch.exp(f) if use_log else 1 + f


def clip_bias(field, magnitude):
    """adv_bias.py:337-356."""
    return 1 + torch.clamp(field - 1, -magnitude, magnitude)


# --------------------------------------------------------------------------------------
# AdvAffine numerics
# --------------------------------------------------------------------------------------
def affine_theta(param, cfg, spatial_dims):
    """Bounded parameters -> (N,d,d+1) matrices.  adv_affine.py:210-273 (Q8)."""
    p = F.hardtanh(param)
    if spatial_dims == 2:
        a = p[:, 0] * cfg["rot"] * math.pi
        sx = 1 + p[:, 1] * cfg["scale_x"]
        sy = 1 + p[:, 2] * cfg["scale_y"]
        row0 = torch.stack([sx * torch.cos(a), sy * (-torch.sin(a)), p[:, 3] * cfg["shift_x"]], dim=-1)
        row1 = torch.stack([sx * torch.sin(a), sy * torch.cos(a), p[:, 4] * cfg["shift_y"]], dim=-1)
        return torch.stack([row0, row1], dim=1)
    n = param.shape[0]
    O = torch.zeros(n, dtype=torch.float32, device=param.device)
    I = torch.ones(n, dtype=torch.float32, device=param.device)

    def mat(rows):
        return torch.stack([torch.stack(r, dim=-1) for r in rows], dim=1)

    T = mat([[I, O, O, p[:, 6] * cfg["shift_x"]], [O, I, O, p[:, 7] * cfg["shift_y"]],
             [O, O, I, p[:, 8] * cfg["shift_z"]], [O, O, O, I]])
    S = mat([[1 + p[:, 3] * cfg["scale_x"], O, O, O], [O, 1 + p[:, 4] * cfg["scale_y"], O, O],
             [O, O, 1 + p[:, 5] * cfg["scale_z"], O], [O, O, O, I]])
    ph = p[:, 0] * cfg["rot_x"] * math.pi
    th = p[:, 1] * cfg["rot_y"] * math.pi
    ps = p[:, 2] * cfg["rot_z"] * math.pi
    c, s = torch.cos, torch.sin
    R = mat([[c(th) * c(ps), -c(ph) * s(ps) + s(ph) * s(th) * c(ps), s(ph) * s(ps) + c(ph) * s(th) * c(ps), O],
             [c(th) * s(ps), c(ph) * c(ps) + s(ph) * s(th) * s(ps), -s(ph) * c(ps) + c(ph) * s(th) * s(ps), O],
             [-s(th), s(ph) * c(th), c(ph) * c(th), O],
             [O, O, O, I]])
    return torch.matmul(T, torch.matmul(R, S))[:, :3, :4]


def affine_inverse(theta):
    """adv_affine.py:316-324."""
    d = theta.shape[1]
    homo = torch.eye(d + 1, dtype=torch.float32, device=theta.device).repeat(theta.shape[0], 1, 1)
    homo[:, :d] = theta
    return homo.inverse()[:, :d, :]


def affine_warp(data, theta, interp, padding_mode):
    """adv_affine.py:289-314."""
    grid = F.affine_grid(theta, data.size(), align_corners=True)
    return _warp_with_padding(data, grid, interp, padding_mode)


# --------------------------------------------------------------------------------------
# consistency loss  (advchain/common/loss.py)
# --------------------------------------------------------------------------------------
def _contour_term(inp, tgt, mask):
    """One-class edge-map MSE.  common/loss.py:102-220 with ignore_background=False,
    one_hot_target=False (as called from loss.py:76-77).  Q14: in 3D conv_x and conv_y share the
    kernel (hp (x) h^T) (x) h and conv_z uses (h (x) h^T) (x) hp."""
    d = inp.dim() - 2
    m = mask[:, :1]
    if d == 2:
        kx = torch.tensor([[1., 0., -1.], [2., 0., -2.], [1., 0., -1.]]).reshape(1, 1, 3, 3)
        ky = torch.tensor([[1., 2., 1.], [0., 0., 0.], [-1., -2., -1.]]).reshape(1, 1, 3, 3)
        kx, ky = kx.to(inp.device), ky.to(inp.device)
        gxp, gyp = F.conv2d(inp, kx, padding=1) * m, F.conv2d(inp, ky, padding=1) * m
        gxt, gyt = F.conv2d(tgt, kx, padding=1) * m, F.conv2d(tgt, ky, padding=1) * m
        return 0.5 * (F.mse_loss(gxp, gxt) + F.mse_loss(gyp, gyt))
    h = np.array([[1, 2, 1]])
    hp = np.array([[1, 0, -1]])
    gx = (hp * h.T).reshape(3, 3, 1) * h
    gz = (h * h.T).reshape(3, 3, 1) * hp
    kx = torch.from_numpy(gx.reshape(1, 1, 3, 3, 3)).float().to(inp.device)
    ky = kx
    kz = torch.from_numpy(gz.reshape(1, 1, 3, 3, 3)).float().to(inp.device)
    terms = 0.
    for k in (kx, ky, kz):
        terms = terms + F.mse_loss(F.conv3d(inp, k, padding=1) * m, F.conv3d(tgt, k, padding=1) * m)
    return 1 / 3 * terms


def consistency_loss(output, reference, divergence_types=("mse", "contour"),
                     divergence_weights=(1.0, 0.5), mask=None, is_gt=False):
    """common/loss.py:8-87 with scales=[0] (adv_compose_solver.py:231), class_weights=None (Q13)."""
    num_classes = reference.size(1)
    if mask is None:
        mask = torch.ones_like(output).float()
    dist = 0.
    for kind, weight in zip(divergence_types, divergence_weights):
        if kind == "kl":
            # loss.py:223-249
            if not is_gt:
                p = F.softmax(reference, dim=1)
                log_p = F.log_softmax(reference, dim=1)
            else:
                p = torch.where(reference == 0, 1e-8, 1 - 1e-8)
                log_p = torch.log(p)
            plogp = torch.sum(mask * (p * log_p), dim=1)
            plogq = torch.sum(mask * (p * F.log_softmax(output, dim=1)), dim=1)
            loss = torch.mean(plogp - plogq)
        elif kind == "mse":
            # loss.py:55-64
            tgt = reference if is_gt else torch.softmax(reference, dim=1)
            inp = torch.softmax(output, dim=1)
            loss = F.mse_loss(inp * mask, tgt * mask) / (torch.numel(mask) / num_classes)
        elif kind == "contour":
            # loss.py:65-79
            tgt = reference if is_gt else torch.softmax(reference, dim=1)
            inp = torch.softmax(output, dim=1)
            loss, cnt = 0., 0
            for i in range(1, num_classes):
                cnt += 1
                loss = loss + _contour_term(inp[:, [i]], tgt[:, [i]], mask)
            if cnt > 0:
                loss = loss / cnt
        else:
            raise NotImplementedError(kind)
        dist = dist + weight * loss
    return dist / 1.0


# --------------------------------------------------------------------------------------
# model-state helpers  (advchain/common/utils.py:114-173, common/layers.py)
# --------------------------------------------------------------------------------------
class OracleFixableDropout(torch.nn.Module):
    """common/layers.py:5-63 (Fixable2DDropout / Fixable3DDropout; ``dims`` picks dropout2d / dropout3d): the seed of
    the last mask is kept and replayed in training mode when ``lazy_load`` is set."""

    def __init__(self, p=0.5, dims=2, inplace=False, lazy_load=False, training=True):
        super().__init__()
        assert 0 <= p <= 1
        self.p, self.dims, self.inplace = p, dims, inplace
        self.seed = None
        self.lazy_load = lazy_load
        self.training = training

    def forward(self, X):
        if self.training and self.lazy_load and self.seed is not None:      # layers.py:20-26
            seed = self.seed
        else:
            seed = torch.seed()                                              # layers.py:27-30
        self.seed = seed
        torch.manual_seed(seed)
        fn = F.dropout2d if self.dims == 2 else F.dropout3d
        return fn(X, p=self.p, training=self.training, inplace=self.inplace)


def _flip_fixable_dropout(model):
    """common/utils.py:139-141,164-167: every Fixable*Dropout gets ``lazy_load = not lazy_load`` (duck-typed on the
    attribute, so the reference's, the product's and the oracle's layer classes are all recognised)."""
    for _, mod in model.named_modules():
        if hasattr(mod, "lazy_load") and hasattr(mod, "seed"):
            mod.lazy_load = not mod.lazy_load


@contextlib.contextmanager
def frozen_bn_stats(model):
    """common/utils.py:114-147 (BatchNorm running-stat tracking off inside the block; Fixable*Dropout.lazy_load is
    flipped on entry and flipped back on exit)."""
    saved = {}
    for name, mod in model.named_modules():
        if isinstance(mod, (torch.nn.BatchNorm2d, torch.nn.BatchNorm3d)):
            saved[name] = mod.track_running_stats
            mod.track_running_stats = False
    _flip_fixable_dropout(model)
    try:
        yield
    finally:
        for name, mod in model.named_modules():
            if name in saved:
                mod.track_running_stats = saved[name]
        _flip_fixable_dropout(model)


@contextlib.contextmanager
def fixed_dropout(model):
    """common/utils.py:149-173 (_fix_dropout)."""
    _flip_fixable_dropout(model)
    try:
        yield
    finally:
        _flip_fixable_dropout(model)


# --------------------------------------------------------------------------------------
# transform plug-ins (stateful, same lifecycle as the reference's classes)
# --------------------------------------------------------------------------------------
class _OracleTransform(object):
    """Parameter lifecycle of adv_transformation_base.py:5-189."""
    name = "base"
    geometric = 0

    def __init__(self, spatial_dims, config_dict, power_iteration=False):
        assert spatial_dims in (2, 3)
        assert len(config_dict["data_size"]) == spatial_dims + 2
        self.spatial_dims = spatial_dims
        self.config_dict = config_dict
        self.data_size = config_dict["data_size"]
        self.power_iteration = power_iteration
        self.param = None
        self.is_training = False
        self.diff = None
        self.step_size = 1

    def get_name(self):
        return self.name

    def is_geometric(self):
        return self.geometric

    def get_step_size(self):
        return self.step_size

    def set_parameters(self, param):
        self.param = param.detach().clone()

    def eval(self):
        if self.is_training:
            self.param.requires_grad = False
            self.is_training = False

    def _as_leaf(self, p):
        self.is_training = True
        self.param = torch.nn.Parameter(p, requires_grad=True)

    def predict_forward(self, data, interp=None, padding_mode=None):
        return data

    def predict_backward(self, data, interp=None, padding_mode=None):
        return data

    def backward(self, data, interp=None, padding_mode=None):
        return data


class OracleNoise(_OracleTransform):
    """adv_noise.py:10-117."""
    name = "noise"

    def __init__(self, spatial_dims, config_dict, power_iteration=False, ignore_values=None):
        super().__init__(spatial_dims, config_dict, power_iteration)
        self.epsilon = config_dict["epsilon"]
        self.xi = config_dict["xi"]
        self.ignore_values = ignore_values

    def init_parameters(self):
        self.param = unit_normalize(torch.randn(*self.data_size, dtype=torch.float32))
        return self.param

    def train(self):
        if self.param is None:
            self.init_parameters()
        self._as_leaf(unit_normalize(self.param) if self.power_iteration else self.param)

    def forward(self, data, **kwargs):
        if self.param is None:
            self.init_parameters()
        scale = self.xi if (self.power_iteration and self.is_training) else self.epsilon
        out = data + scale * self.param
        if self.ignore_values is not None:
            out[(abs(data - self.ignore_values) < 1e-8).detach().clone()] = self.ignore_values
        self.diff = out - data
        return out

    def optimize_parameters(self, step_size=None):
        step_size = self.step_size if step_size is None else step_size
        g = unit_normalize(self.param.grad)
        self.param = g.detach() if self.power_iteration else (self.param + step_size * g.detach()).detach()
        return self.param

    def rescale_parameters(self):
        self.param = unit_normalize(self.param)


class OracleBias(_OracleTransform):
    """adv_bias.py:50-380."""
    name = "bias"

    def __init__(self, spatial_dims, config_dict, power_iteration=False, ignore_values=None):
        super().__init__(spatial_dims, config_dict, power_iteration)
        self.ignore_values = ignore_values
        self.xi = 1e-6
        self.epsilon = config_dict["epsilon"]

    def init_parameters(self):
        c = self.config_dict
        self.magnitude = c["epsilon"]
        assert 0 <= self.magnitude < 1
        self.use_log = c["space"] == "log"
        self.geom = BiasGeometry(c["data_size"], c["control_point_spacing"], c["downscale"],
                                 c["interpolation_order"])
        self.low, self.high = -np.inf, np.inf
        mode = c["init_mode"]
        if mode == "gaussian":
            self.param = torch.ones(*self.geom.cp_grid, dtype=torch.float32).normal_(mean=0, std=0.5)
        elif mode == "random":
            if self.use_log:
                self.low, self.high = np.log(1 - self.magnitude), np.log(1 + self.magnitude)
            else:
                self.low, self.high = -self.magnitude, self.magnitude
            self.param = torch.rand(*self.geom.cp_grid, dtype=torch.float32) * (self.high - self.low) + self.low
        elif mode == "identity":
            self.param = torch.zeros(*self.geom.cp_grid, dtype=torch.float32)
        else:
            raise NotImplementedError
        self.bias_field = clip_bias(bias_field(self.param, self.geom, self.use_log), self.magnitude)
        return self.param

    def train(self):
        self._as_leaf(unit_normalize(self.param.data) if self.power_iteration else self.param.data)

    def compute_smoothed_bias(self, cpoint=None):
        return bias_field(self.param if cpoint is None else cpoint, self.geom, self.use_log)

    def forward(self, data, **kwargs):
        if self.param is None:
            self.init_parameters()
        cp = self.xi * self.param if (self.power_iteration and self.is_training) else self.param
        field = bias_field(cp, self.geom, self.use_log)
        if field.size(1) < data.size(1):
            field = field.expand(data.size())
        field = clip_bias(field, self.magnitude)
        self.bias_field = field
        self.diff = field
        if self.ignore_values is not None:
            assert isinstance(self.ignore_values, float)
            mask = (abs(data - self.ignore_values) < 1e-8).detach().clone()
            out = data * field
            out[mask] = self.ignore_values
            return out
        return field * data

    def optimize_parameters(self, step_size=0.3):
        g = unit_normalize(self.param.grad)
        if self.power_iteration:
            self.param = g.clone().detach()
        else:
            self.param = (self.param + step_size * g.detach()).clone().detach()
        return self.param

    def rescale_parameters(self):
        self.param = torch.clamp(self.param, self.low, self.high)


def one_ulp_jitter(seed=0):
    """Field hook: moves every value of a sampling grid by exactly one fp32 ulp, up or down at random, as a CONSTANT
    offset (gradients flow through unchanged).  A parity tolerance on a quantity that this perturbation moves by x in
    the reference / oracle itself cannot be tighter than x: that is rounding sensitivity, not a kernel difference."""
    def hook(q):
        g = torch.Generator().manual_seed(seed + q.numel())
        up = torch.rand(q.shape, generator=g) < 0.5
        d = q.detach()
        target = torch.where(up, torch.full_like(d, float("inf")), torch.full_like(d, float("-inf")))
        return q + (torch.nextafter(d, target) - d)
    return hook


def uniform_jitter(amplitude, seed=0):
    """Field hook: constant offsets drawn uniformly from [-amplitude, amplitude] (normalised grid units) -- the
    reference's sensitivity to a field that differs from its own by `amplitude` (e.g. another implementation's
    rounding), see one_ulp_jitter."""
    def hook(q):
        g = torch.Generator().manual_seed(seed + q.numel())
        return q + (torch.rand(q.shape, generator=g) * 2 - 1) * amplitude
    return hook


class OracleMorph(_OracleTransform):
    """adv_morph.py:204-564."""
    name = "morph"
    geometric = 1

    def __init__(self, spatial_dims, config_dict, power_iteration=False, image_padding_mode="zeros"):
        super().__init__(spatial_dims, config_dict, power_iteration)
        self.image_padding_mode = image_padding_mode
        self.xi = 0.5
        self._read_config()
        # Q10: the constructor resets the interp modes AFTER init_config read the dict
        self.forward_interp = "bilinear"
        self.backward_interp = "bilinear"

    def _read_config(self):
        c = self.config_dict
        self.epsilon = c["epsilon"]
        self.vector_size = c["vector_size"]
        if "forward_interp" in c:
            self.forward_interp = c["forward_interp"]
        if "backward_interp" in c:
            self.backward_interp = c["backward_interp"]

    def init_parameters(self):
        self._read_config()
        v = torch.rand(self.data_size[0], self.spatial_dims, *self.vector_size) * 2 - 1
        self.param = unit_normalize(v)
        return self.param

    def train(self):
        if self.param is None:
            self.init_parameters()
        self._as_leaf(unit_normalize(self.param) if self.power_iteration else self.param)

    field_hook = None   # oracle-only: callable(field) -> field applied to every DemonsCompose output (sensitivity runs)

    def _field(self, sign):
        scale = self.xi if (self.power_iteration and self.is_training) else self.epsilon
        q = demons_compose(sign * scale * self.param, self.data_size[2:])
        return q if self.field_hook is None else self.field_hook(q)


    def _warp(self, data, dxy, interp, padding_mode):
        if padding_mode is None:
            padding_mode = self.image_padding_mode
        return _warp_with_padding(data, _to_sampler_layout(dxy), interp, padding_mode)

    def forward(self, data, interp=None, padding_mode=None):
        if self.param is None:
            self.init_parameters()
        interp = self.forward_interp if interp is None else interp
        dxy = torch.clamp(self._field(+1), -1, 1)
        out = self._warp(data, dxy, interp, padding_mode)
        self.diff = out - data
        self.dxy = dxy
        return out

    def backward(self, data, interp=None, padding_mode=None):
        interp = self.backward_interp if interp is None else interp
        return self._warp(data, self._field(-1), interp, padding_mode)

    predict_forward = forward
    predict_backward = backward

    def optimize_parameters(self, step_size=None):
        g = unit_normalize(self.param.grad)
        self.param = g.detach() if self.power_iteration else (self.param + step_size * g.detach()).detach()
        return self.param

    def rescale_parameters(self):
        self.param = unit_normalize(self.param)


class OracleAffine(_OracleTransform):
    """adv_affine.py:13-330."""
    name = "affine"
    geometric = 1

    def __init__(self, spatial_dims, config_dict, power_iteration=False, image_padding_mode="zeros"):
        super().__init__(spatial_dims, config_dict, power_iteration)
        self.image_padding_mode = image_padding_mode
        self.xi = 1e-6
        self._read_config()
        self.forward_interp = "bilinear"
        self.backward_interp = "bilinear"

    def _read_config(self):
        c = self.config_dict
        if "forward_interp" in c:
            self.forward_interp = c["forward_interp"]
        if "backward_interp" in c:
            self.backward_interp = c["backward_interp"]

    def init_parameters(self):
        self._read_config()
        n = 5 if self.spatial_dims == 2 else 9
        self.param = F.hardtanh(2 * torch.rand(self.data_size[0], n, dtype=torch.float32) - 1)
        return self.param

    def train(self):
        self._as_leaf(self.param.sign() if self.power_iteration else self.param)

    def forward(self, data, interp=None, padding_mode=None):
        if self.param is None:
            self.init_parameters()
        interp = self.forward_interp if interp is None else interp
        p = self.xi * self.param if (self.power_iteration and self.is_training) else self.param
        self.affine_matrix = affine_theta(p, self.config_dict, self.spatial_dims)
        # Q9: any caller-supplied padding_mode is replaced by the constructor's one
        out = affine_warp(data, self.affine_matrix, interp, self.image_padding_mode)
        self.diff = data - out
        return out

    def backward(self, data, interp=None, padding_mode=None):
        assert self.param is not None
        interp = self.backward_interp if interp is None else interp
        return affine_warp(data, affine_inverse(self.affine_matrix), interp, self.image_padding_mode)

    predict_forward = forward
    predict_backward = backward

    def optimize_parameters(self, step_size=None):
        s = self.param.grad.sign().detach()
        self.param = s if self.power_iteration else (self.param + step_size * s).detach()
        return self.param

    def rescale_parameters(self):
        return self.param


# --------------------------------------------------------------------------------------
# solver  (advchain/augmentor/adv_compose_solver.py)
# --------------------------------------------------------------------------------------
class OracleSolver(object):
    """adv_compose_solver.py:11-538, same public control flow; ``trace`` (oracle-only) records
    per-step dist / raw grads / params for the teacher-forced parity protocol."""

    def __init__(self, chain, divergence_types=("mse", "contour"), divergence_weights=(1.0, 0.5),
                 if_norm_image=False, min_intensity=None, max_intensity=None, is_gt=False):
        self.chain = list(chain)
        self.divergence_types = list(divergence_types)
        self.divergence_weights = list(divergence_weights)
        self.if_norm_image = if_norm_image
        self.min_intensity, self.max_intensity = min_intensity, max_intensity
        self.is_gt = is_gt
        self.trace = []

    # -- chain application: adv_compose_solver.py:148-219
    def forward(self, data, chain=None):
        chain = self.chain if chain is None else chain
        t = data.detach().clone()
        for tr in chain:
            t = tr.forward(t)
        if self.if_norm_image:
            lo = torch.min(data) if self.min_intensity is None else self.min_intensity
            hi = torch.max(data) if self.max_intensity is None else self.max_intensity
            t = torch.clamp(t, lo, hi)
        return t

    def predict_forward(self, data, chain=None):
        for tr in (self.chain if chain is None else chain):
            data = tr.predict_forward(data)
        return data

    def predict_backward(self, data, chain=None):
        for tr in reversed(self.chain if chain is None else chain):
            data = tr.predict_backward(data)
        return data

    def has_geometric(self, chain=None):
        return sum(tr.is_geometric() for tr in (self.chain if chain is None else chain)) > 0

    def loss_fn(self, pred, reference, mask=None):
        return consistency_loss(pred, reference, self.divergence_types, self.divergence_weights,
                                mask=mask, is_gt=self.is_gt)

    def get_init_output(self, model, data):
        with torch.no_grad():
            with frozen_bn_stats(model):
                return model(data)

    # -- adv_compose_solver.py:281-287 (Q18: value only, zero gradient)
    def anatomy_misoverlap(self, anatomy):
        rec = self.predict_backward(self.predict_forward(anatomy))
        rec[rec >= 0.5] = 1
        rec[rec < 0.5] = 0
        return F.mse_loss(rec, anatomy)

    # -- adv_compose_solver.py:479-500
    def init_random_transformation(self, lazy_load=False, anatomy=None, tol=5e-4):
        for tr in self.chain:
            if lazy_load:
                if tr.param is None:
                    tr.init_parameters()
            else:
                tr.init_parameters()
            if tr.is_geometric() == 1 and anatomy is not None:
                tries = 0
                while self.anatomy_misoverlap(anatomy) > tol:
                    tr.init_parameters()
                    tries += 1
                    if tries > 10:
                        break

    def _masked_dist(self, pred_back, init_output):
        ones = torch.ones_like(init_output)
        m = self.predict_backward(self.predict_forward(ones))
        m[m != 0] = 1
        return self.loss_fn(pred_back, init_output, m), m

    # -- adv_compose_solver.py:289-405
    def optimizing_transform(self, model, data, init_output, optimize_flags, n_iter, step_sizes,
                             anatomy=None, anatomy_reg_weight=50, tol=5e-4):
        stop = not (n_iter > 0)
        i_iter, one_time = 0, n_iter
        transforms = []
        while not stop:
            model.zero_grad()
            i_iter += 1
            for flag, tr in zip(optimize_flags, self.chain):
                if flag:
                    tr.train()
            params_in = [None if tr.param is None else tr.param.detach().clone() for tr in self.chain]
            aug = self.forward(data.detach().clone())
            with frozen_bn_stats(model):
                out = model(aug)
            if self.has_geometric():
                back = self.predict_backward(out)
                dist, _ = self._masked_dist(back, init_output)
                if anatomy is not None and abs(anatomy_reg_weight) > 1e-32:
                    dist = dist + anatomy_reg_weight * self.anatomy_misoverlap(anatomy)
            else:
                dist = self.loss_fn(out, init_output.detach())
            rec = {"dist": float(dist.detach()), "params_in": params_in}
            if not (torch.isnan(dist) or torch.isinf(dist)):
                dist.backward()
                rec["grads"] = [None if (tr.param is None or tr.param.grad is None)
                                else tr.param.grad.detach().clone() for tr in self.chain]
                for flag, tr in zip(optimize_flags, self.chain):
                    if flag:
                        # reference quirk (adv_compose_solver.py:349-364): the index ``i_tr`` is
                        # never advanced, so EVERY transform is stepped with step_sizes[0]
                        try:
                            step = step_sizes[0]
                        except Exception:
                            step = tr.get_step_size()
                        tr.optimize_parameters(step_size=step)
            rec["params_out"] = [None if tr.param is None else tr.param.detach().clone() for tr in self.chain]
            self.trace.append(rec)
            model.zero_grad()
            if i_iter == n_iter:
                transforms = []
                for flag, tr in zip(optimize_flags, self.chain):
                    if flag:
                        tr.rescale_parameters()
                        tr.eval()
                    transforms.append(tr)
                if self.has_geometric(transforms) and anatomy is not None and abs(anatomy_reg_weight) > 1e-32:
                    if abs(self.anatomy_misoverlap(anatomy)) <= tol:
                        stop = True
                    else:
                        if i_iter >= 3 * one_time:
                            stop = True
                            self.init_random_transformation(anatomy=anatomy, tol=tol)
                        elif i_iter == 2 * one_time:
                            self.init_random_transformation(anatomy=anatomy, tol=tol)
                            n_iter += one_time
                        else:
                            n_iter += 1
                        for flag, tr in zip(optimize_flags, self.chain):
                            if flag:
                                tr.train()
                        transforms.append(tr)  # reference quirk: last transform appended again (399)
                else:
                    stop = True
        return transforms

    # -- adv_compose_solver.py:236-279
    def calc_adv_consistency_loss(self, data, model, init_output, chain=None):
        chain = self.chain if chain is None else chain
        for tr in chain:
            tr.eval()
        adv = self.forward(data, chain)
        old = model.training
        model.train()
        with fixed_dropout(model):                       # adv_compose_solver.py:256-259
            adv_out = model(adv.detach().clone())
        if self.has_geometric(chain):
            ones = torch.ones_like(init_output)
            m = self.predict_backward(self.predict_forward(ones, chain), chain)
            back = self.predict_backward(adv_out, chain)
            m[m != 0] = 1
            dist = self.loss_fn(back, init_output.detach(), m)
        else:
            back = adv_out
            dist = self.loss_fn(adv_out, init_output.detach())
        model.train(old)
        return dist, adv, adv_out, back

    # -- adv_compose_solver.py:43-146
    def adversarial_training(self, data, model, optimize_flags=None, init_output=None, lazy_load=False,
                             power_iteration=False, n_iter=1, step_sizes=None, anatomy_mask_images=None,
                             anatomy_reg_weight=50, volume_preserve_tolerance=5e-4):
        k = len(self.chain)
        if optimize_flags is None:
            optimize_flags = [n_iter > 0] * k
        if isinstance(power_iteration, bool):
            pis = [power_iteration] * k
        elif isinstance(power_iteration, list):
            pis = power_iteration
        else:
            assert power_iteration == "smart"
            pis = [tr.get_name() == "noise" for tr in self.chain]
        for tr, pi in zip(self.chain, pis):
            tr.power_iteration = pi
        if step_sizes is None:
            step_sizes = [1] * k
        elif isinstance(step_sizes, (float, int)):
            step_sizes = [step_sizes] * k
        if init_output is None:
            init_output = self.get_init_output(model, data)
        self.init_random_transformation(lazy_load, anatomy_mask_images, volume_preserve_tolerance)
        if n_iter >= 1:
            self.chain = self.optimizing_transform(model, data, init_output, optimize_flags, n_iter,
                                                   step_sizes, anatomy_mask_images, anatomy_reg_weight,
                                                   volume_preserve_tolerance)
        dist, adv, adv_out, back = self.calc_adv_consistency_loss(data.detach().clone(), model, init_output)
        self.init_output, self.adv_data, self.adv_predict = init_output, adv, adv_out
        self.warped_back_adv_output = back
        return dist
