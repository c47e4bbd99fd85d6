"""Shared test helpers: golden-fixture loading and oracle construction (tests only)."""
import contextlib
import json
import math
import os

import numpy as np
import torch
import torch.nn.functional as F

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Fixture(object):
    def __init__(self, name):
        self._d = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.keys = set(self._d.files)

    def __contains__(self, k):
        return k in self.keys

    def t(self, k, device="cpu"):
        return torch.from_numpy(np.array(self._d[k])).to(device)

    def f(self, k):
        return float(self._d[k])

    def json(self, k="meta"):
        return json.loads(str(self._d[k]))

    def arr(self, k):
        return np.array(self._d[k])


def smooth_data(n, c, dims, seed):
    """Same generator as oracle/make_golden.py (needed where a fixture stores only the seed)."""
    g = torch.Generator().manual_seed(seed)
    coarse = torch.rand(n, c, *([6] * len(dims)), generator=g)
    mode = "bilinear" if len(dims) == 2 else "trilinear"
    return F.interpolate(coarse, size=tuple(dims), mode=mode, align_corners=True).clamp(0, 1).contiguous()


def rand(shape, seed, lo=-1.0, hi=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(*shape, generator=g) * (hi - lo) + lo


def make_model(spatial_dims, k=4, device="cpu"):
    """Conv(1,k,3,1,1) with closed-form weights."""
    if spatial_dims == 2:
        m = torch.nn.Conv2d(1, k, 3, 1, 1)
        a, b = torch.meshgrid(torch.arange(3.).double(), torch.arange(3.).double(), indexing="ij")
        w = torch.stack([0.1 * (q + 1) * ((a - 1) + 2 * (b - 1)) + 0.05 for q in range(k)])[:, None]
    else:
        m = torch.nn.Conv3d(1, k, 3, 1, 1)
        a, b, c = torch.meshgrid(*([torch.arange(3.).double()] * 3), indexing="ij")
        w = torch.stack([0.05 * (q + 1) * ((a - 1) + 2 * (b - 1) - (c - 1)) + 0.02 for q in range(k)])[:, None]
    m.weight.data = w.float().contiguous()
    m.bias.data = torch.tensor([0.01 * q for q in range(k)])
    return m.eval().to(device)


def oracle_chain(spec):
    """Oracle transforms from a fixture's chain spec."""
    from oracle import advchain_oracle as O
    chain = []
    for s in spec:
        cfg, kw = s["config"], s.get("kwargs", {})
        sd = len(cfg["data_size"]) - 2
        cls = {"noise": O.OracleNoise, "bias": O.OracleBias, "morph": O.OracleMorph, "affine": O.OracleAffine}[s["name"]]
        chain.append(cls(sd, cfg, **kw))
    return chain


def maxdiff(a, b):
    return float((a.double() - b.double()).abs().max())


# ------------------------------------------------------------------ BatchNorm / Fixable*Dropout test models (a11)
def device_independent(dropout_cls):
    """Subclass of a Fixable{2,3}DDropout class (the reference's, the product's or the oracle's) that keeps the parent's
    seed / lazy_load logic but draws the (N, C, 1, ...) feature mask on the CPU generator, so that a CPU reference run
    and a GPU product run see the same mask for the same seed (the CUDA and CPU generators differ)."""
    class DeviceIndependentDropout(dropout_cls):
        def forward(self, X):
            ones = torch.ones(X.shape[0], X.shape[1], *([1] * (X.dim() - 2)))
            return X * super().forward(ones).to(X.device)
    return DeviceIndependentDropout


@contextlib.contextmanager
def counted_torch_seed(start=1000):
    """``torch.seed()`` draws from the OS: inside this block it returns start+1, start+2, ... instead, so that runs of
    models with Fixable*Dropout layers are reproducible (same patch in oracle/make_golden.py and in the tests)."""
    orig, state = torch.seed, [start]

    def fake():
        state[0] += 1
        torch.manual_seed(state[0])
        return state[0]
    torch.seed = fake
    try:
        yield
    finally:
        torch.seed = orig


def bn_dropout_model(sd, kind, make_dropout):
    """Conv + BatchNorm + Fixable dropout models with closed-form weights.  ``make_dropout(p)`` builds the dropout
    layer (so the same architecture is built on the reference's, the product's and the oracle's layer classes).
      'train': Conv(1,4,3) -> BN(4) -> Dropout(0.3) -> Conv(4,4,1), left in train() mode (the SSL use of README:175-278)
      'eval' : Conv(1,4,3) -> BN(4) -> Dropout(0.1) -> Softmax(dim=1) in eval() mode with non-trivial running
               statistics (the 3D notebook's toy model, example/adv_chain_data_generation_cardiac_2D_3D.ipynb cell 26)"""
    conv = torch.nn.Conv2d if sd == 2 else torch.nn.Conv3d
    bn = torch.nn.BatchNorm2d if sd == 2 else torch.nn.BatchNorm3d
    first = make_model(sd)
    norm = bn(4)
    c = torch.arange(4.)
    norm.weight.data = 1 + 0.1 * c
    norm.bias.data = 0.05 * c
    norm.running_mean.data = 0.1 * c - 0.1
    norm.running_var.data = 1 + 0.2 * c
    if kind == "train":
        last = conv(4, 4, 1)
        w = torch.tensor([[0.25 * math.cos(1.3 * q + 0.7 * k) for k in range(4)] for q in range(4)])
        last.weight.data = w.reshape(4, 4, *([1] * sd)).contiguous()
        last.bias.data = 0.02 * c
        return torch.nn.Sequential(first, norm, make_dropout(0.3), last).train()
    return torch.nn.Sequential(first, norm, make_dropout(0.1), torch.nn.Softmax(dim=1)).eval()


def make_model_multi(spatial_dims, in_ch, k=4, device="cpu"):
    """Conv(in_ch,k,3,1,1): channel c of the single-channel closed-form kernel scaled by (1 - 0.3 c)."""
    base = make_model(spatial_dims, k)
    conv = (torch.nn.Conv2d if spatial_dims == 2 else torch.nn.Conv3d)(in_ch, k, 3, 1, 1)
    conv.weight.data = torch.cat([base.weight.data * (1 - 0.3 * c) for c in range(in_ch)], dim=1).contiguous()
    conv.bias.data = base.bias.data.clone()
    return conv.eval().to(device)


def fixture_model(meta, dropout_classes, device="cpu"):
    """The model a g6 fixture was generated with; ``dropout_classes`` = (2D class, 3D class) of the implementation under
    test (product layers, or the oracle's)."""
    sd = meta["spatial_dims"]
    if meta.get("model"):
        cls = device_independent(dropout_classes[sd - 2])
        return bn_dropout_model(sd, meta["model"], lambda p: cls(p)).to(device)
    if meta.get("in_ch", 1) > 1:
        return make_model_multi(sd, meta["in_ch"], device=device)
    return make_model(sd, device=device)


# ------------------------------------------------------------------ realistic-size one-step fixtures (g6l_*)
def notebook_configs(dims, batch, names, morph_div8=False):
    """Transform configurations of the reference's notebooks at a given size (the same table as
    bench.transform_configs) -- shared by oracle/make_golden.py (g6l fixtures) and the tests that replay them."""
    sd = len(dims)
    ds = [batch, 1] + list(dims)
    out = []
    for nm in names:
        if nm == "noise":
            out.append((nm, dict(epsilon=1.0, xi=1e-6, data_size=ds)))
        elif nm == "bias":
            out.append((nm, dict(epsilon=0.3, control_point_spacing=[s // 2 for s in dims],
                                 downscale=2 if sd == 2 else 4, data_size=ds, interpolation_order=3,
                                 init_mode="random", space="log")))
        elif nm == "morph":
            if morph_div8:
                vs = [s // 8 for s in dims]
            elif sd == 2:
                vs = [s // 16 for s in dims]
            else:
                vs = [dims[0] // 16, dims[1] // 16, dims[2] // 2]
            out.append((nm, dict(epsilon=1.5, data_size=ds, vector_size=vs)))
        else:
            if sd == 2:
                out.append((nm, dict(rot=30.0 / 180, scale_x=0.2, scale_y=0.2, shift_x=0.1, shift_y=0.1, data_size=ds)))
            else:
                out.append((nm, dict(rot_x=10.0 / 180, rot_y=10.0 / 180, rot_z=10.0 / 180, scale_x=0.1, scale_y=0.1,
                                     scale_z=0.1, shift_x=0.1, shift_y=0.1, shift_z=0.1, data_size=ds)))
    return out


def anatomy_blob(n, dims):
    """The 0/1 anatomy mask of the g6l cases that carry one (an ellipsoid of half the extent along every axis, the same for
    every sample): shared by oracle/make_golden.py and the tests that replay the case."""
    axes = torch.meshgrid([torch.linspace(-1, 1, s) for s in dims], indexing="ij")
    blob = (sum(a ** 2 for a in axes) <= 0.5 ** 2 * len(dims)).float()
    return blob[None, None].repeat(n, 1, *([1] * len(dims))).contiguous()


def seeded_init_param(name, shape, seed):
    """Initial parameters of a g6l case, from a seed (the fixtures do not store them): bias well inside its clip range,
    affine at 60 % of its bounds, noise / morph on the unit L2 sphere per sample (what init_parameters() produces)."""
    r = rand(tuple(shape), seed)
    if name == "bias":
        return 0.1 * r
    if name == "affine":
        return 0.6 * r
    flat = r.reshape(r.shape[0], -1)
    return (flat / (flat.norm(dim=1, keepdim=True) + 1e-20)).reshape(r.shape).contiguous()


SAMPLE_STRIDE = 61      # tensors above SAMPLE_FULL elements are stored as every 61st element + float64 moments
SAMPLE_FULL = 16384     # (8192 until round 5; 16384 keeps cfg-5's 3 x 20 x 20 x 10 velocity gradient whole.  Readers go by key)


def sampled_record(t):
    """What a g6l fixture keeps of a tensor: everything (small) or a strided sample plus float64 sum / |sum| / sum of
    squares / max of the whole tensor (large)."""
    flat = t.detach().reshape(-1).double().cpu()
    if flat.numel() <= SAMPLE_FULL:
        return dict(full=t.detach().float().cpu())
    return dict(samples=flat[::SAMPLE_STRIDE].float(), moments=torch.stack([flat.sum(), flat.abs().sum(), (flat ** 2).sum(),
                                                                           flat.abs().max()]))


def compare_sampled(fx, key, t, tol_abs):
    """Max |difference| of tensor `t` against the record `key` of fixture `fx` (full tensor, or the strided sample and
    the moments: sums divided by the element count, i.e. mean differences, and the maximum)."""
    flat = t.detach().reshape(-1).double().cpu()
    if key + "__full" in fx:
        return maxdiff(t.detach().cpu(), fx.t(key + "__full"))
    err = float((flat[::SAMPLE_STRIDE] - fx.t(key + "__samples").double()).abs().max())
    m = fx.t(key + "__moments").double()
    n = flat.numel()
    mine = torch.stack([flat.sum(), flat.abs().sum(), (flat ** 2).sum(), flat.abs().max()])
    # moments: mean signed / absolute difference (sums over the count) and the maximum
    err = max(err, float((mine[0] - m[0]).abs()) / n, float((mine[1] - m[1]).abs()) / n, float((mine[3] - m[3]).abs()))
    return err
