"""Host-side construction of band tables for the tensor-product interpolation kernels.

A band table describes, per spatial axis, a small linear map (g coefficients -> S samples, at most
8 contiguous non-zeros per sample).  Two maps are used on the hot path:

* linear upsampling with ``align_corners=False`` (ATen ``UpSample.h:259-311``) -- the velocity
  upsample of ``adv_morph.py:464`` and the last stage of the bias field (``adv_bias.py:316-327``);
* the cubic B-spline synthesis ``conv_transpose(cp, kernel) -> crop`` of ``adv_bias.py:12-49,293-307``
  in closed form, composed with the upsample into ONE per-axis matrix.

This is geometry set-up (O(S) numbers per axis, done once per transform), not the data path.
"""
import numpy as np
import torch

BAND_MAX = 8


def linear_upsample_matrix(in_size, out_size, scale_factor=None):
    """(out_size, in_size) matrix of F.interpolate(mode=linear, align_corners=False), fp32 index
    arithmetic exactly as ATen (area_pixel_compute_scale / compute_source_index / guard_index_and_lambda)."""
    if scale_factor is not None and scale_factor > 0:
        scale = np.float32(1.0 / float(scale_factor))
    else:
        scale = np.float32(in_size) / np.float32(out_size)
    dst = np.arange(out_size, dtype=np.float32)
    src = scale * (dst + np.float32(0.5)) - np.float32(0.5)
    src = np.where(src < 0, np.float32(0), src).astype(np.float32)
    i0 = np.minimum(src.astype(np.int64), in_size - 1)
    lam1 = np.clip(src - i0.astype(np.float32), 0, 1).astype(np.float32)
    lam0 = (np.float32(1) - lam1).astype(np.float32)
    i1 = i0 + (i0 < in_size - 1)
    M = np.zeros((out_size, in_size), dtype=np.float64)
    rows = np.arange(out_size)
    np.add.at(M, (rows, i0), lam0.astype(np.float64))
    np.add.at(M, (rows, i1), lam1.astype(np.float64))
    return M


def bspline_kernel_1d(spacing, order, variant):
    """1-D factor of the reference's separable B-spline window (adv_bias.py:12-49).

    variant '2d': round i pads by i*s (window ends up zero-padded to 10s+3 for order 3);
    variant '3d': every round pads s-1 (length 4s-3).  Returned in float64."""
    s = int(spacing)
    k = np.ones(s, dtype=np.float64)
    ones = np.ones(s, dtype=np.float64)
    for i in range(1, order + 1):
        pad = i * s if variant == "2d" else s - 1
        k = np.convolve(np.pad(k, pad), ones, mode="valid") / s
    return k


def bspline_synthesis_matrix(n_cp, spacing, order, variant, crop_start, crop_end):
    """(h, n_cp) matrix of conv_transpose(cp, kernel, stride=s, padding=(len-1)//2) followed by the crop
    [s + crop_start : -s - crop_end]  (adv_bias.py:293-307, 370-371)."""
    s = int(spacing)
    k = bspline_kernel_1d(s, order, variant)
    K = len(k)
    pad = int((K - 1) / 2)
    full_len = (n_cp - 1) * s - 2 * pad + K
    full = np.zeros((full_len, n_cp), dtype=np.float64)
    for i in range(n_cp):
        for t in range(K):
            o = i * s - pad + t
            if 0 <= o < full_len:
                full[o, i] += k[t]
    lo = s + int(crop_start)
    hi = full_len - s - int(crop_end)
    return full[lo:hi]


class BandTables(object):
    """Device band tables for a tensor-product map (g0,g1[,g2]) -> (S0,S1[,S2])."""

    def __init__(self, mats, device):
        mats = [np.asarray(m, dtype=np.float64) for m in mats]
        self.ndim = len(mats)
        assert self.ndim in (2, 3)
        if self.ndim == 2:  # trivial leading axis so the kernels always see 3 axes
            mats = [np.ones((1, 1))] + mats
        ints, floats = [], []
        self.S, self.g, self.B = [], [], []
        for M in mats:
            S, g = M.shape
            nz = M != 0
            first = np.where(nz.any(1), nz.argmax(1), 0)
            last = np.where(nz.any(1), g - 1 - nz[:, ::-1].argmax(1), 0)
            B = int(max(1, (last - first + 1).max()))
            if B > BAND_MAX:
                raise NotImplementedError("interpolation band %d exceeds the kernel limit %d" % (B, BAND_MAX))
            B = min(B, g)
            start = np.clip(np.minimum(first, g - B), 0, None).astype(np.int32)
            w = np.zeros((S, B), dtype=np.float32)
            for j in range(B):
                w[:, j] = M[np.arange(S), start + j]
            lo = np.zeros(g, dtype=np.int32)
            hi = np.zeros(g, dtype=np.int32)
            for k in range(g):
                touched = np.nonzero((start <= k) & (k < start + B))[0]
                if len(touched):
                    lo[k], hi[k] = touched[0], touched[-1] + 1
            ints += [start, lo, hi]
            floats.append(w.reshape(-1))
            self.S.append(S)
            self.g.append(g)
            self.B.append(B)
        self.itab = torch.from_numpy(np.concatenate(ints).astype(np.int32)).to(device)
        self.ftab = torch.from_numpy(np.concatenate(floats).astype(np.float32)).to(device)
        self.mats = mats  # kept on the host for tests / debugging
        # the densified bands of the innermost axis (advchain_band_reduce_rows_dense): wd[k][j] = weight of input lo[k] + j
        # for coefficient k -- what the adjoint kernels used to rebuild per workgroup from (start, lo, hi, w)
        start, lo, hi = ints[-3], ints[-2], ints[-1]
        w = floats[-1].reshape(self.S[-1], self.B[-1])
        WB = int(max(1, (hi - lo).max()))
        wd = np.zeros((self.g[-1], WB), dtype=np.float32)
        for k in range(self.g[-1]):
            for j in range(int(hi[k] - lo[k])):
                s2 = int(lo[k]) + j
                b = k - int(start[s2])
                if 0 <= b < self.B[-1]:
                    wd[k, j] = w[s2, b]
        self.dense_inner = (torch.from_numpy(wd.reshape(-1)).to(device), torch.from_numpy(lo.astype(np.int32)).to(device), WB)

    @property
    def coef_dims(self):
        return self.g[3 - self.ndim:]

    @property
    def full_dims(self):
        return self.S[3 - self.ndim:]


_TABLE_CACHE = {}


def cached_tables(key, device, build):
    """Band tables are a pure function of the geometry: build (and upload) them once per (geometry, device) -- the
    transforms re-initialise their parameters on every solver call, and each upload is a synchronous host-to-device
    copy that drains the launch queue."""
    k = (key, str(device))
    t = _TABLE_CACHE.get(k)
    if t is None:
        if len(_TABLE_CACHE) > 64:
            _TABLE_CACHE.clear()
        t = _TABLE_CACHE[k] = build()
    return t


def upsample_tables(low_dims, full_dims, device, scale_factors=None):
    key = ("upsample", tuple(low_dims), tuple(full_dims), None if scale_factors is None else tuple(scale_factors))
    return cached_tables(key, device, lambda: _upsample_tables(low_dims, full_dims, device, scale_factors))


def _upsample_tables(low_dims, full_dims, device, scale_factors=None):
    mats = []
    for a, (l, f) in enumerate(zip(low_dims, full_dims)):
        sf = None if scale_factors is None else scale_factors[a]
        mats.append(linear_upsample_matrix(l, f, sf))
    return BandTables(mats, device)


def gaussian_taps(sigma=1.0, gaussian_ks=5):
    """Window length of the reference's Gaussian (adv_morph.py:393-398): the caller's `gaussian_ks` unless the rule
    2 * int(4 sigma + 0.5) + 1 is larger (sigma = 1, gaussian_ks = 5 -> 9 taps, Q3; sigma = 0.3 -> the 5 of gaussian_ks)."""
    return max(int(gaussian_ks), 2 * int(4 * float(sigma) + 0.5) + 1)


def gaussian_weights_1d(sigma=1.0, taps=None):
    """1-D factor of the normalised k^d window of adv_morph.py:391-428 (sigma=1 -> 9 taps), centred on (k - 1) / 2."""
    k = 2 * int(4 * sigma + 0.5) + 1 if taps is None else int(taps)
    t = np.arange(k, dtype=np.float64) - (k - 1) / 2.0
    w = np.exp(-t * t / (2.0 * sigma * sigma))
    return (w / w.sum()).tolist()
