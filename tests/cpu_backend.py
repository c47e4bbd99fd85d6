"""TEST-ONLY operator backend: the signatures of ``advchain_amd.ops`` implemented with the CPU oracle's
torch ops, so that the product's HOST logic (transform classes, solver control flow, band tables, loss
normalisers, batch sharding) can be exercised on a machine without a GPU and under ``gloo``.

It is installed by the ``cpu_ops`` fixture (monkeypatching module attributes) and never ships: the product
itself has no CPU path and raises on CPU tensors."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import advchain_oracle as O


def _planar_to_last(grid):
    d = grid.dim() - 2
    return grid.permute(0, *range(2, 2 + d), 1)


def _rider(ride, g, mode, padding_mode, nonzero):
    with torch.no_grad():
        r = F.grid_sample(ride, g.detach(), mode=mode, padding_mode=padding_mode, align_corners=True)
        return (r != 0).to(r.dtype) if nonzero else r


def grid_sample(inp, grid, interp="bilinear", padding_mode="zeros", clamp_grid=False, ride=None, ride_nonzero=False):
    if clamp_grid:
        grid = torch.clamp(grid, -1, 1)
    mode = interp if interp in ("nearest", "bicubic") else "bilinear"
    g = _planar_to_last(grid)
    out = F.grid_sample(inp, g, mode=mode, padding_mode=padding_mode, align_corners=True)
    return out if ride is None else (out, _rider(ride, g, mode, padding_mode, ride_nonzero))


def affine_warp(inp, theta, interp="bilinear", padding_mode="zeros", ride=None, ride_nonzero=False):
    mode = interp if interp in ("nearest", "bicubic") else "bilinear"
    g = F.affine_grid(theta, inp.size(), align_corners=True)
    out = F.grid_sample(inp, g, mode=mode, padding_mode=padding_mode, align_corners=True)
    return out if ride is None else (out, _rider(ride, g, mode, padding_mode, ride_nonzero))


def affine_theta(param, cfg, param_scale, nd):
    keys = (["rot", "scale_x", "scale_y", "shift_x", "shift_y"] if nd == 2 else
            ["rot_x", "rot_y", "rot_z", "scale_x", "scale_y", "scale_z", "shift_x", "shift_y", "shift_z"])
    th = O.affine_theta(param_scale * param, dict(zip(keys, cfg)), nd)
    return th, O.affine_inverse(th)


def axpy(x, y, a):
    return x + a * y


def sign_axpy(base, x, a, gate=None, old=None):
    sg = torch.sign(x.detach())
    return a * sg if base is None else base.detach() + a * sg


def nonzero_mask(x):
    return (x.detach() != 0).to(x.dtype)


def normalized_axpy(base, x, step=1.0, gate=None, old=None):
    u = O.unit_normalize(x.detach())
    return step * u if base is None else base.detach() + step * u


def _tp_eval(coef, tables):
    """Dense evaluation of the band tables (host copies): (N,C,g...) -> (N,C,S...)."""
    mats = [torch.from_numpy(np.asarray(m)).to(coef.dtype) for m in tables.mats[3 - tables.ndim:]]
    if tables.ndim == 2:
        return torch.einsum("ncij,xi,yj->ncxy", coef, mats[0], mats[1])
    return torch.einsum("ncijk,xi,yj,zk->ncxyz", coef, mats[0], mats[1], mats[2])


def bias_apply(cp, data, tables, eps, use_log=True, cp_scale=1.0):
    L = cp_scale * _tp_eval(cp, tables)
    e = torch.exp(L) if use_log else 1 + L
    field = 1 + torch.clamp(e - 1, -eps, eps)
    return data * field, field.detach()


def bias_field_only(cp, tables, eps, use_log=True, cp_scale=1.0):
    return bias_apply(cp.detach(), torch.ones(1), tables, eps, use_log, cp_scale)[1]


def demons_field(vel, scale, tables, nsteps_rule, reduce_sumsq=None, opts=None):
    """adv_morph.py:454-491 without the final clamp; the 3D step-count norm may be reduced across ranks.
    opts = (num_steps, smooth_iter, sigma, positions_only[, taps]) as in advchain_amd.ops._DemonsField."""
    n, smooth_iter, sigma, pos_only = tuple(opts)[:4] if opts is not None else (8, 1, 1.0, False)
    taps = opts[4] if (opts is not None and len(opts) > 4) else 2 * int(4 * sigma + 0.5) + 1
    if taps != 9 and not pos_only:
        raise NotImplementedError("9-tap window only")
    dims = tuple(tables.full_dims)
    d = len(dims)
    base = O.identity_grid(vel.shape[0], dims)
    u = scale * vel
    for _ in range(smooth_iter):
        u = O.gaussian_smooth(u, sigma=sigma, kernel_size=taps)
    u = F.interpolate(u, size=dims, mode="bilinear" if d == 2 else "trilinear", align_corners=False)
    if nsteps_rule:
        ss = (u.detach().double() ** 2).sum().float().reshape(1)
        if reduce_sumsq is not None:
            ss = reduce_sumsq(ss)
        norm = float(ss.sqrt())
        while norm / (2.0 ** n) > 0.5:
            n += 1
    phi0 = base + u / (2.0 ** n)
    phi = phi0
    for _ in range(n):
        phi = O.compose_fields(phi, phi)
    if pos_only:
        return (phi - phi0) + base
    composed = O.compose_fields(base, (phi - phi0) + base)
    return O.gaussian_smooth(composed - base, sigma=sigma, kernel_size=taps) + base


def demons_field_pair(vel, scale, tables, nsteps_rule, reduce_sumsq=None, opts=None):
    return (demons_field(vel, scale, tables, nsteps_rule, reduce_sumsq, opts),
            demons_field(vel, -scale, tables, nsteps_rule, reduce_sumsq, opts))


def gauss_smooth(x, sigma=1.0, taps=None):
    return O.gaussian_smooth(x, sigma=sigma, kernel_size=5 if taps is None else taps)


def upsample_field(coef, tables, scale):
    dims = tuple(tables.full_dims)
    d = len(dims)
    return O.identity_grid(coef.shape[0], dims) + scale * F.interpolate(coef, size=dims, mode="bilinear" if d == 2 else "trilinear",
                                                                         align_corners=False)


def consistency_sums(pred, ref, mask, coef, ref_is_prob=False, want_edges=True):
    """[S_mse, S_edgeA, S_edgeB, S_kl] raw sums as the HIP kernels define them (advchain_amd/csrc/loss.hip)."""
    K = pred.shape[1]
    d = pred.dim() - 2
    P = torch.softmax(pred, dim=1)
    T = ref if ref_is_prob else torch.softmax(ref, dim=1)
    m = torch.ones_like(pred[:, :1]) if mask is None else mask
    s0 = ((P * m - T * m) ** 2).sum()
    sa = sb = torch.zeros(())
    if want_edges and K > 1:
        m0 = m[:, :1]
        if d == 2:
            ka = torch.tensor([[1., 0., -1.], [2., 0., -2.], [1., 0., -1.]]).reshape(1, 1, 3, 3)
            kb = torch.tensor([[1., 2., 1.], [0., 0., 0.], [-1., -2., -1.]]).reshape(1, 1, 3, 3)
            conv = F.conv2d
        else:
            h = torch.tensor([1., 2., 1.])
            hp = torch.tensor([1., 0., -1.])
            ka = torch.einsum("i,j,k->ijk", h, hp, h).reshape(1, 1, 3, 3, 3)
            kb = torch.einsum("i,j,k->ijk", h, h, hp).reshape(1, 1, 3, 3, 3)
            conv = F.conv3d
        D = (P - T)[:, 1:]
        Dr = D.reshape(-1, 1, *D.shape[2:])
        mr = m0.unsqueeze(1).expand(-1, K - 1, *m0.shape[1:]).reshape(-1, 1, *m0.shape[2:])
        sa = ((conv(Dr, ka, padding=1) * mr) ** 2).sum()
        sb = ((conv(Dr, kb, padding=1) * mr) ** 2).sum()
    coef = list(coef) + [0.0] * (4 - len(coef))
    skl = torch.zeros(())
    if coef[3] != 0.0:      # 'kl' row (loss.py:223-249)
        p = torch.where(ref == 0, 1e-8, 1 - 1e-8) if ref_is_prob else T
        log_p = torch.log(p) if ref_is_prob else F.log_softmax(ref, dim=1)
        skl = (m * (p * log_p)).sum() - (m * (p * F.log_softmax(pred, dim=1))).sum()
    sums = torch.stack([s0, sa, sb, skl])
    return torch.dot(sums, torch.tensor(coef, dtype=sums.dtype)), sums.detach()


PATCHED = ["grid_sample", "affine_warp", "affine_theta", "axpy", "normalized_axpy", "sign_axpy", "nonzero_mask", "bias_apply", "bias_field_only",
           "demons_field", "demons_field_pair", "gauss_smooth", "upsample_field", "consistency_sums"]


def install(monkeypatch):
    from advchain_amd import ops
    import tests.cpu_backend as me
    for name in PATCHED:
        monkeypatch.setattr(ops, name, getattr(me, name))
