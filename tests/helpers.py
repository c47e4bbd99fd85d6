"""Shared test helpers: golden-fixture loading and oracle construction (tests only)."""
import json
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
