"""Autograd-composable operators backed by the HIP kernels (C ABI via ctypes).

Every operator requires CUDA(ROCm) fp32 tensors and the built ``libadvchain_hip.so``; there is no
PyTorch / CPU fallback -- a CPU tensor or a missing library raises.  PyTorch is used for device
memory (``torch.empty``), the current stream and the autograd tape only.
"""
import ctypes
import functools
import os

import torch

from . import _lib
from .bands import gaussian_weights_1d

_INTERP = {"bilinear": 0, "trilinear": 0, "linear": 0, "nearest": 1, "bicubic": 2}
_PAD = {"zeros": 0, "border": 1, "reflection": 2}
_GAUSS9 = _lib.float_array(gaussian_weights_1d(1.0))
_GAUSS9_CACHE = {}


GENERIC_MAX_TAPS = 129


def gauss9(sigma, taps=None):
    """The tap weights of the Gaussian for `sigma` (ctypes array).  The reference sizes its window as
    max(gaussian_ks, 2 * int(4 sigma + 0.5) + 1) taps (adv_morph.py:393-398; `taps`, default the rule alone): 9 for
    0.875 <= sigma < 1.125 with the default gaussian_ks = 5 -- the fast kernels; any other window runs
    advchain_gauss_axis_generic (raw_gauss; no fused prologue / epilogue there)."""
    sigma = float(sigma)
    if taps is None:
        taps = 2 * int(4 * sigma + 0.5) + 1
    taps = int(taps)
    if sigma == 1.0 and taps == 9:
        return _GAUSS9
    w = _GAUSS9_CACHE.get((sigma, taps))
    if w is None:
        if not sigma > 0.0:
            raise ValueError("Gaussian smoothing needs sigma > 0, got %r" % (sigma,))
        if taps % 2 == 0:
            # (the reference's conv then pads k // 2 on both sides and returns a field one voxel larger per axis)
            raise NotImplementedError("Gaussian smoothing with an even window (%d taps) is not implemented" % taps)
        if taps > GENERIC_MAX_TAPS:
            raise NotImplementedError("Gaussian smoothing with sigma=%g needs a %d-tap window (at most %d)"
                                      % (sigma, taps, GENERIC_MAX_TAPS))
        if len(_GAUSS9_CACHE) > 16:
            _GAUSS9_CACHE.clear()
        w = _GAUSS9_CACHE[(sigma, taps)] = _lib.float_array(gaussian_weights_1d(sigma, taps))
    return w


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    # torch.cuda.current_stream() builds a Stream object through five Python layers (~10 us; ~300 calls per solver
    # call): ask the C layer for the raw handle of the current stream of the current device instead
    if _raw_stream is not None and _raw_device is not None:
        return ctypes.c_void_p(_raw_stream(_raw_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


_STREAM_OBJ = {}


def _stream_obj():
    """The current stream of the current device as a torch Stream object (Event.record() without an argument builds one
    through five Python layers on every call): cached per device, rebuilt when the raw handle has changed."""
    if _raw_stream is None or _raw_device is None:
        return torch.cuda.current_stream()
    dev = _raw_device()
    s = _STREAM_OBJ.get(dev)
    if s is None or s.cuda_stream != _raw_stream(dev):
        s = _STREAM_OBJ[dev] = torch.cuda.current_stream()
    return s


def _dev(t, name="tensor"):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.AdvchainHipError("%s must be a CUDA/ROCm tensor: the advchain_amd kernels have no CPU path" % name)
    if t.dtype != torch.float32:
        raise _lib.AdvchainHipError("%s must be float32, got %s" % (name, t.dtype))
    return t if t.is_contiguous() else t.contiguous()


def _same_device(*tensors):
    """All operands of one launch must live on one GPU (raw pointers carry no device)."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError("Expected all tensors to be on the same device, but found at least two devices, "
                               "%s and %s!" % (dev, t.device))


def _on_tensor_device(fn):
    """Run `fn` with the first tensor argument's GPU as the current device: `_stream()` hands the kernels the current
    stream of the CURRENT device, so an operator called on cuda:1 tensors while cuda:0 is current would otherwise
    launch on device 0's stream against device-1 pointers.  The common case (already current) costs one C call.
    (Autograd runs a Function's backward on the device thread of its forward: only entry points need the guard.)"""
    @functools.wraps(fn)
    def guarded(*args, **kwargs):
        for a in args:
            if isinstance(a, (tuple, list)) and a and isinstance(a[0], torch.Tensor):
                a = a[0]
            if isinstance(a, torch.Tensor):
                if a.is_cuda:
                    cur = _raw_device() if _raw_device is not None else torch.cuda.current_device()
                    if a.device.index != cur:
                        with torch.cuda.device(a.device):
                            return fn(*args, **kwargs)
                break
        return fn(*args, **kwargs)
    return guarded


def interp_code(interp):
    if interp not in _INTERP:
        raise NotImplementedError("interpolation mode %r is not implemented by the HIP sampler "
                                  "(supported: bilinear/trilinear, nearest, bicubic (2D))" % (interp,))
    return _INTERP[interp]


def pad_code(padding_mode):
    if padding_mode not in _PAD:
        raise NotImplementedError("padding_mode %r" % (padding_mode,))
    return _PAD[padding_mode]


# ------------------------------------------------------------------------------------------------
# raw (non-autograd) wrappers: one C-ABI call each
# ------------------------------------------------------------------------------------------------
def _hint_bits(disp):
    """Displacement estimate (voxels) -> the hint bits 8..15 of an int argument (0 = unknown); see include/advchain_hip.h."""
    if disp is None or not disp == disp or disp < 0:
        return 0
    return min(int(disp) + 1, 255) << 8


def _fine_bits(disp):
    """Displacement estimate in 1/1024 voxel (at least 1 when known, 0 = unknown): bits 8.. of a chain hint."""
    if disp is None or not disp == disp or disp < 0:
        return 0
    return max(1, min(int(disp * 1024.0) + 1, (1 << 22) - 1))


def raw_grid_sample_fwd(inp, grid, interp, padding, clamp_grid, disp_hint=None):
    N, C = inp.shape[:2]
    nd = inp.dim() - 2
    odims = grid.shape[2:]
    out = torch.empty((N, C) + tuple(odims), device=inp.device, dtype=torch.float32)
    _lib.check(_lib.load().advchain_grid_sample_fwd(_ptr(inp), _ptr(grid), _ptr(out), N, C, nd,
                                                    _lib.dims_array(inp.shape[2:]), _lib.dims_array(odims),
                                                    interp, padding, int(clamp_grid) | _hint_bits(disp_hint), _stream()),
               "grid_sample_fwd")
    return out


def raw_grid_sample_fwd_ride(inp, grid, ride, interp, padding, clamp_grid, nonzero, disp_hint=None):
    """raw_grid_sample_fwd for `inp` and a one-channel rider through the same grid (one launch in 2D)."""
    N, C = inp.shape[:2]
    nd = inp.dim() - 2
    odims = tuple(grid.shape[2:])
    out = torch.empty((N, C) + odims, device=inp.device, dtype=torch.float32)
    rout = torch.empty((N, 1) + odims, device=inp.device, dtype=torch.float32)
    _lib.check(_lib.load().advchain_grid_sample_fwd_ride(_ptr(inp), _ptr(grid), _ptr(out), _ptr(ride), _ptr(rout), N, C, nd,
                                                         _lib.dims_array(inp.shape[2:]), _lib.dims_array(odims), interp, padding,
                                                         int(clamp_grid) | _hint_bits(disp_hint), int(bool(nonzero)), _stream()),
               "grid_sample_fwd_ride")
    return out, rout


DISP_SLOTS = 4096     # ADVCHAIN_DISP_SLOTS of include/advchain_hip.h
ADAPTIVE_HALO = os.environ.get("ADVCHAIN_NO_ADAPTIVE_HALO") is None   # measure the displacement in forward and size the backward halos from it
PAIR_FIELDS = os.environ.get("ADVCHAIN_NO_PAIR_FIELDS") is None       # a solver step integrates field(+v) and field(-v) as one batch
TILED_SCATTER = True  # LDS-tiled owner-computes scatter (False: global-atomic kernels; for A/B tests)
FUSED_LOSS = True     # the consistency loss straight from the logits (no P / D intermediates); False: A/B tests
FUSE_2D = True        # the leading sub-pixel squarings of a 2D chain in one launch (expo_fused2d.hip); False: A/B tests
COMPOSITE = True      # a paired 2D DemonsCompose direction as ONE C call (demons_compose.cpp: same launches); False: A/B tests
RIDE_MASK = True      # the solver's validity mask rides through the data's warps (one launch for both); False: A/B tests
RIDE_INTERPS = ("bilinear", "trilinear", "linear", "nearest")


def set_deterministic(on):
    """Process-wide: route the one backward formulation whose bits depend on arrival order (the window scatter's float-atomic
    flush) through its 64-bit fixed-point twin (include/advchain_hip.h: advchain_set_deterministic).  The solver sets it at
    the start of every call from its `deterministic` attribute; two solvers with different settings in one process are fine as
    long as their calls do not interleave (a workspace is sized when it is allocated, for the mode of that moment)."""
    _lib.load().advchain_set_deterministic(1 if on else 0)


def is_deterministic():
    return bool(_lib.load().advchain_get_deterministic())


def _scatter_workspace(N, dims, device):
    n = _lib.load().advchain_scatter_workspace(N, len(dims), _lib.dims_array(dims))
    return torch.empty(n, device=device, dtype=torch.int32)


def raw_grid_sample_bwd(gout, inp, grid, interp, padding, clamp_grid, need_gin, need_ggrid, halo=0):
    N, C = inp.shape[:2]
    nd = inp.dim() - 2
    tiled = TILED_SCATTER and need_gin
    ws = _scatter_workspace(N, inp.shape[2:], inp.device) if tiled else None
    gin = (torch.empty_like(inp) if tiled else torch.zeros_like(inp)) if need_gin else None
    ggrid = torch.empty_like(grid) if need_ggrid else None
    _lib.check(_lib.load().advchain_grid_sample_bwd(_ptr(gout), _ptr(inp), _ptr(grid), _ptr(gin), _ptr(ggrid), _ptr(ws),
                                                    N, C, nd, _lib.dims_array(inp.shape[2:]),
                                                    _lib.dims_array(grid.shape[2:]), interp, padding, int(clamp_grid),
                                                    int(halo), _stream()), "grid_sample_bwd")
    return gin, ggrid


def raw_compose_self_fwd(phi, phi0=None, final_mode=0, disp_out=None, disp_hint=None):
    """phi o phi; `disp_out` (DISP_SLOTS zero-initialised floats) max-accumulates the displacement of the result.
    `disp_hint`: the caller's estimate of phi's displacement in voxels (picks the forward kernel; results do not depend on it)."""
    N = phi.shape[0]
    nd = phi.dim() - 2
    out = torch.empty_like(phi)
    _lib.check(_lib.load().advchain_compose_self_fwd(_ptr(phi), _ptr(out), _ptr(phi0), N, nd,
                                                     _lib.dims_array(phi.shape[2:]), final_mode | _hint_bits(disp_hint), _ptr(disp_out),
                                                     _stream()), "compose_self_fwd")
    return out


def raw_compose_self_bwd(gout, phi, ws=None, chain=False, halo=0):
    """chain=True: ``gout`` is the result of the previous call that used the same workspace ``ws``."""
    N = phi.shape[0]
    nd = phi.dim() - 2
    if TILED_SCATTER and ws is None:
        ws, chain = _scatter_workspace(N, phi.shape[2:], phi.device), False
    gphi = torch.empty_like(phi) if ws is not None else torch.zeros_like(phi)
    _lib.check(_lib.load().advchain_compose_self_bwd(_ptr(gout), _ptr(phi), _ptr(gphi), _ptr(ws), int(bool(chain)), int(halo), N, nd,
                                                     _lib.dims_array(phi.shape[2:]), _stream()), "compose_self_bwd")
    return gphi


def raw_max_displacement(phi, out=None):
    """max |sampling position - own voxel| of a deformation `phi` (N,d,...) in voxels -> 1-element device tensor (`out`: a
    ZEROED 1-element tensor to accumulate into)."""
    if out is None:
        out = torch.zeros(1, device=phi.device, dtype=torch.float32)
    _lib.check(_lib.load().advchain_max_displacement(_ptr(phi), _ptr(out), phi.shape[0], phi.dim() - 2,
                                                     _lib.dims_array(phi.shape[2:]), _stream()), "max_displacement")
    return out


def raw_slot_rows_max(slots, reset=False, out=None):
    """Row maxima of a (rows, slots) displacement accumulator (torch.max(dim=1) semantics incl. NaN); `reset` zeroes the
    accumulator behind the read (a persistent buffer then needs no zero-fill launch per chain)."""
    if out is None:
        out = torch.empty(slots.shape[0], device=slots.device, dtype=torch.float32)
    _lib.check(_lib.load().advchain_slot_rows_max(_ptr(slots), _ptr(out), slots.shape[0], slots.shape[1], int(bool(reset)),
                                                  _stream()), "slot_rows_max")
    return out


_PERSISTENT = {}


def _persistent_zeros(tag, shape, device):
    """A zero-initialised device buffer that lives for the process, keyed by (tag, shape, device, stream): accumulators
    whose CONSUMER kernel zeroes them again (advchain_slot_rows_max / advchain_consistency_finish with reset) -- one
    torch.zeros at first use instead of a fill launch per call.  Stream order makes the reuse safe: the consumer of call
    k runs before the producers of call k + 1 on the same stream."""
    stream = _raw_stream(_raw_device()) if (_raw_stream and _raw_device) else torch.cuda.current_stream().cuda_stream
    key = (tag, tuple(shape), str(device), stream)
    plan = _PLAN
    if plan is not None and plan.is_frozen:
        # a capture keeps its accumulators with its plan.  freeze() made the ones the recorded calls asked for (zero-filled
        # THEN, outside the capture: a torch.zeros met inside a capture is a fill node replayed with every call, for buffers
        # whose consumers leave them zeroed anyway); one that was not asked for before is made here, inside the capture
        key = key[:3]
        buf = plan.persistent.get(key)
        if buf is None:
            buf = plan.persistent[key] = torch.zeros(shape, device=device, dtype=torch.float32)
        return buf
    if plan is not None and plan.recording:
        plan.wanted_zeros.add(key[:3])
    buf = _PERSISTENT.get(key)
    if buf is None:
        if len(_PERSISTENT) > 64:
            _PERSISTENT.clear()
        buf = _PERSISTENT[key] = torch.zeros(shape, device=device, dtype=torch.float32)
    return buf


def _forget_persistent(buf):
    """Drop a persistent accumulator whose producer ran but whose consumer did not (an exception in between): it may hold
    partial sums, and the next user must start from a fresh zero-filled buffer."""
    stores = [_PERSISTENT] + ([_PLAN.persistent] if _PLAN is not None else [])
    for store in stores:
        for k in [k for k, v in store.items() if v is buf]:
            del store[k]


def raw_gauss_small_pair(x, scale, adjoint=False, weights=None):
    """Gaussian of the low-resolution planes of a paired field: forward (N,d,..) -> (2N,d,..) = [G(s x); G(-s x)], adjoint
    (2N,d,..) -> (N,d,..) = G(s x[:N]) - G(s x[N:]).  None when the planes are too large for the one-launch kernel."""
    if x[0, 0].numel() > 4096 or not x.is_contiguous() or (weights is not None and len(weights) != 9):
        return None
    nd = x.dim() - 2
    n_in = x.shape[0]
    n_out = n_in // 2 if adjoint else 2 * n_in
    out = torch.empty((n_out,) + tuple(x.shape[1:]), device=x.device, dtype=torch.float32)
    planes = (n_out if adjoint else n_in) * x.shape[1]
    _lib.check(_lib.load().advchain_gauss_small_pair(_ptr(x), _ptr(out), planes, nd, _lib.dims_array(x.shape[2:]),
                                                     weights or _GAUSS9, float(scale), int(bool(adjoint)), _stream()),
               "gauss_small_pair")
    return out


def raw_gauss_small_pair_into(x, out, scale, weights=None):
    """raw_gauss_small_pair(x, scale) (forward form) into a buffer the caller allocated."""
    _lib.check(_lib.load().advchain_gauss_small_pair(_ptr(x), _ptr(out), x.shape[0] * x.shape[1], x.dim() - 2,
                                                     _lib.dims_array(x.shape[2:]), weights or _GAUSS9, float(scale), 0, _stream()),
               "gauss_small_pair")


@_on_tensor_device
def sign_axpy(base, x, a, gate=None, old=None):
    """base + a * sign(x) (base may be None); no autograd (parameter updates, adv_affine.py:186-195).  gate / old: see
    normalized_axpy."""
    x = _dev(x.detach(), "x")
    base = None if base is None else _dev(base.detach(), "base")
    gate, old = _gate_args(gate, old, x)
    out = torch.empty_like(x)
    _lib.check(_lib.load().advchain_sign_axpy(_ptr(base), _ptr(x), _ptr(out), float(a), x.numel(), _ptr(gate), _ptr(old),
                                              _stream()), "sign_axpy")
    return out


@_on_tensor_device
def nonzero_mask(x):
    """(x != 0) as float32 (the validity mask, adv_compose_solver.py:266-268); no autograd."""
    x = _dev(x.detach(), "x")
    out = torch.empty_like(x)
    _lib.check(_lib.load().advchain_nonzero_mask(_ptr(x), _ptr(out), x.numel(), _stream()), "nonzero_mask")
    return out


_ONES = {}


def cached_ones(shape, device):
    """A read-only tensor of ones (the input of the validity-mask warps): filled once per shape, never written."""
    key = (tuple(shape), str(device))
    plan = _PLAN
    store = plan.persistent if (plan is not None and plan.is_frozen) else _ONES
    t = store.get(("ones",) + key) if store is not _ONES else _ONES.get(key)
    if t is None:
        if store is _ONES:
            if len(_ONES) > 16:
                _ONES.clear()
            t = _ONES[key] = torch.ones(shape, device=device, dtype=torch.float32)
        else:
            # read-only: the tensor the recorded calls filled serves the capture as well (the plan keeps it alive); only a
            # shape nobody asked for before is filled under the capture (a fill node replayed with every call)
            t = _ONES.get(key)
            if t is None:
                t = torch.ones(shape, device=device, dtype=torch.float32)
            store[("ones",) + key] = t
    return t


def squaring_halo(disp, d):
    """Displacement bound for the backward of one squaring from the MEASURED displacement of its input (voxels).
    Negative = exact (the gather-form adjoint then needs no overflow list): the measurement uses the kernels' own
    arithmetic, so `disp < H` guarantees every sample is within H voxels.  Larger displacements size the tile halo of
    the scatter kernels (a hint: anything beyond it goes through their overflow list)."""
    if not disp == disp:
        return 0                 # NaN field: nothing to tune
    if disp < 0.999:
        return -1
    if d == 3:       # exact bounds of 2..4 voxels: owner-computes march (scatter_march.hip); beyond: window scatter
        return _halo_3d(disp)
    if 16 - 0.001 <= disp < 32 - 0.001:
        # the whole-row scatter of a squaring still beats the window scatter here (58 against 79 us at 24 px, no zero fill,
        # deterministic)
        return -24 if disp < 24 - 0.001 else -32
    return _halo_2d(disp)


def _halo_3d(disp):
    """3D above one voxel: exact bounds of 2..4 voxels (owner-computes march scatter); beyond: a hint for the window
    scatter.  (The C ABI also takes exact bounds of 5..8 voxels -- one march launch per channel -- but on the smooth
    fields of the solver the window scatter is faster there: cfg-5 121.6 against 131.7 ms per call, cfg-4 61.4 against
    63.1; on rougher fields it is the other way round, tools/kernel_bench.py: C=4 681 against 2153 us.  Re-measured in round 4
    with the flat march scatter of round 3 in place, same box, bounds 6 / 8 handed to the march: cfg-5 99.0 against 89.2 ms
    per call -- still a loss.)"""
    for h in (2, 3, 4):
        if disp < h - 0.001:
            return -h
    return 8


def _halo_2d(disp):
    """2D: exact bounds of 2 (gather form / whole rows) and 3 .. 16 pixels (whole-row owner-computes scatter); beyond: a
    hint for the window scatter.  The whole-row kernel takes any bound; a finer ladder than powers of two (round 5) costs
    nothing -- a bound of 3 / 6 / 12 / 24 uses the fixed-point resolution of 4 / 8 / 16 / 32 (its scale is the maximum over
    the rows a workgroup visits, so the two agree to that resolution, not bit for bit) -- and visits 2 .. 16 halo rows less
    (0.7 us each at 64 x 2 x 256 x 256): cfg-2 7.39 -> 7.35 ms launch by launch, 7.50 -> 7.44 replayed."""
    for h in (2, 3, 4, 6, 8, 12, 16):
        if disp < h - 0.001:
            return -h
    return 16


def raw_gauss(x, C, pre=0, post=0, scale=1.0, aux=None, weights=None):
    """Separable 9-tap Gaussian over all spatial axes of x (planes = x.shape[0]*x.shape[1]).  `x` may be a pair of
    tensors (the two halves of a batch, e.g. the gradients of a paired field): the fused x+y launch reads both in place,
    any other route concatenates them first.  `weights`: gauss9(sigma), default sigma = 1."""
    w9 = weights or _GAUSS9
    x_hi = None
    if isinstance(x, (tuple, list)):
        x, x_hi = x
        shape = (x.shape[0] + x_hi.shape[0],) + tuple(x.shape[1:])
    else:
        shape = tuple(x.shape)
    nd = x.dim() - 2
    planes = shape[0] * shape[1]
    dims = _lib.dims_array(x.shape[2:])
    axes = [2, 1, 0][:nd]  # innermost first (padded 3-axis numbering)
    lib = _lib.load()
    if len(w9) != 9:      # another window: the plain per-axis kernel (adv_morph.py:393-398 with sigma outside [0.875, 1.125))
        if post != 0 or pre not in (0, 1):
            raise NotImplementedError("the fused prologue / epilogue of the Gaussian exists for the 9-tap window only")
        cur = torch.cat([x, x_hi], 0) if x_hi is not None else x
        cur = cur if cur.is_contiguous() else cur.contiguous()
        for i, ax in enumerate(axes):
            out = torch.empty_like(cur)
            _lib.check(lib.advchain_gauss_axis_generic(_ptr(cur), _ptr(out), planes, nd, dims, ax, w9, len(w9),
                                                       float(scale) if (pre == 1 and i == 0) else 1.0, _stream()),
                       "gauss_axis_generic")
            cur = out
        return cur
    if x_hi is not None and (x[0, 0].numel() <= 4096 or not (x.is_contiguous() and x_hi.is_contiguous())):
        x, x_hi = torch.cat([x, x_hi], 0), None
    if post == 0 and pre in (0, 1) and x[0, 0].numel() <= 4096:     # low-resolution grids: all axes in one launch
        out = torch.empty_like(x)
        _lib.check(lib.advchain_gauss_small(_ptr(x), _ptr(out), planes, nd, dims, w9, pre, float(scale), _stream()),
                   "gauss_small")
        return out
    cur = x
    # x and y in one launch where the shape allows (advchain_gauss_xy); post belongs to the last axis
    out = torch.empty(shape, device=x.device, dtype=torch.float32)
    rc = lib.advchain_gauss_xy(_ptr(x), _ptr(out), _ptr(aux) if (post == 2 and nd == 2) else None, planes, C, nd, dims, w9,
                               pre, post if nd == 2 else 0, float(scale) if pre == 1 else 1.0, _stream(), _ptr(x_hi),
                               x.shape[0] * x.shape[1])
    if rc == 0:
        if nd == 2:
            return out
        out2 = torch.empty_like(out)
        _lib.check(lib.advchain_gauss_axis(_ptr(out), _ptr(out2), _ptr(aux) if post == 2 else None, planes, C, nd, dims, 0,
                                           w9, 0, post, 1.0, _stream()), "gauss_axis")
        return out2
    if rc != -2:
        _lib.check(rc, "gauss_xy")
    if x_hi is not None:
        x = cur = torch.cat([x, x_hi], 0)
    for i, ax in enumerate(axes):
        out = torch.empty_like(x)
        p = pre if i == 0 else 0
        q = post if i == len(axes) - 1 else 0
        _lib.check(lib.advchain_gauss_axis(_ptr(cur), _ptr(out), _ptr(aux) if q == 2 else None, planes, C, nd, dims, ax,
                                           w9, p, q, float(scale) if p == 1 else 1.0, _stream()), "gauss_axis")
        cur = out
    return cur


def raw_tp_interp(coef, tables, C, add_identity=False, scale=1.0, want_out=True, sumsq=None, disp_out=None, out=None):
    planes = coef.shape[0] * coef.shape[1]
    if want_out and out is None:
        out = torch.empty((coef.shape[0], coef.shape[1]) + tuple(tables.full_dims), device=coef.device,
                          dtype=torch.float32)
    _lib.check(_lib.load().advchain_tp_interp_fwd(_ptr(coef), _ptr(out), _ptr(tables.itab), _ptr(tables.ftab),
                                                  _lib.dims_array(tables.S), _lib.dims_array(tables.g),
                                                  _lib.dims_array(tables.B), planes, C, tables.ndim,
                                                  int(add_identity), float(scale), _ptr(sumsq), _ptr(disp_out),
                                                  _stream()),
               "tp_interp_fwd")
    return out


def raw_tp_adjoint(gfull, tables, gfull2=None, scale=1.0):
    """W^T applied along every axis: (N,C,S...) -> (N,C,g...).  First pass may fuse (a - b) * scale."""
    lib = _lib.load()
    N, C = gfull.shape[:2]
    S, g, B = list(tables.S), list(tables.g), list(tables.B)
    Sa, ga, Ba = _lib.dims_array(S), _lib.dims_array(g), _lib.dims_array(B)
    cur, cur2 = gfull, gfull2
    shape = list(S)
    first = True
    for ax in (2, 1, 0):
        if S[ax] == 1 and g[ax] == 1:
            continue
        outer = N * C
        for a in range(ax):
            outer *= shape[a]
        inner = 1
        for a in range(ax + 1, 3):
            inner *= shape[a]
        shape[ax] = g[ax]
        out = torch.empty((N, C) + tuple(shape[3 - tables.ndim:]), device=gfull.device, dtype=torch.float32)
        rc = -2
        if ax == 2 and inner == 1 and getattr(tables, "dense_inner", None) is not None:
            # the full-resolution pass with the bands densified once per table (same sums, same order)
            wd, lo, WB = tables.dense_inner
            rc = lib.advchain_band_reduce_rows_dense(_ptr(cur), _ptr(cur2), _ptr(out), _ptr(wd), _ptr(lo), outer, S[2], g[2], WB,
                                                     float(scale) if first else 1.0, _stream())
            if rc != -2:
                _lib.check(rc, "band_reduce_rows_dense")
        if rc == -2:
            _lib.check(lib.advchain_band_reduce_axis(_ptr(cur), _ptr(cur2), _ptr(out), _ptr(tables.itab), _ptr(tables.ftab),
                                                     Sa, ga, Ba, ax, outer, inner, float(scale) if first else 1.0,
                                                     _stream()), "band_reduce_axis")
        cur, cur2, first = out, None, False
    if first:  # degenerate: nothing to reduce
        cur = (gfull - gfull2 if gfull2 is not None else gfull) * scale
    return cur


def raw_axpy(x, y, a):
    out = torch.empty_like(y)
    _lib.check(_lib.load().advchain_axpy(_ptr(x), _ptr(y), _ptr(out), float(a), y.numel(), _stream()), "axpy")
    return out


def _gate_args(gate, old, like):
    """(gate pointer, old pointer) of the NaN-gated parameter updates: `gate` a 0-dim / 1-element device tensor (the loss of
    the step), `old` what the output falls back to when the gate is not finite."""
    if gate is None:
        return None, None
    gate = _dev(gate.detach().reshape(1), "gate")
    old = _dev(old.detach(), "old")
    if old.shape != like.shape:
        raise RuntimeError("gated update: `old` must have the shape of the result")
    return gate, old


@_on_tensor_device
def normalized_axpy(base, x, step=1.0, gate=None, old=None):
    """base + step * x / (||x||_2 per sample + 1e-20); base may be None.  No autograd (parameter updates).  With `gate` (a
    device scalar) the result is `old` whenever the gate is NaN / inf: the solver's NaN guard without a host read-back."""
    x = _dev(x.detach(), "x")
    base = None if base is None else _dev(base.detach(), "base")
    gate, old = _gate_args(gate, old, x)
    N = x.shape[0]
    M = x.numel() // max(N, 1)
    lib = _lib.load()
    ws = torch.empty(max(1, lib.advchain_norm_workspace(N, M)), device=x.device, dtype=torch.float32)
    out = torch.empty_like(x)
    _lib.check(lib.advchain_norm_axpy_gated(_ptr(base), _ptr(x), _ptr(out), _ptr(ws), float(step), N, M, _ptr(gate), _ptr(old),
                                            _stream()), "norm_axpy")
    return out


# ------------------------------------------------------------------------------------------------
# autograd Functions
# ------------------------------------------------------------------------------------------------
FUSED_UPDATE = os.environ.get("ADVCHAIN_FUSED_UPDATE_OFF") is None   # the parameter updates of an ascent step as ONE launch (advchain_update_multi); False: one per transform (A/B)


@_on_tensor_device
def update_multi(items, gate=None):
    """The parameter updates of one ascent step in one launch.  items: [(base or None, x, step, kind, old)] with kind 0 =
    base + step * x / (||x||_2 per sample + 1e-20), kind 1 = base + step * sign(x); `gate` (a device scalar): a non-finite gate
    keeps `old`.  Returns the new tensors (no autograd: parameter updates)."""
    lib = _lib.load()
    descs = (_lib.UpdateDesc * len(items))()
    outs, keep = [], []
    g = None if gate is None else _dev(gate.detach().reshape(1), "gate")
    for d, (base, x, step, kind, old) in zip(descs, items):
        x = _dev(x.detach(), "x")
        base = None if base is None else _dev(base.detach(), "base")
        old = None if old is None else _dev(old.detach(), "old")
        if g is not None and (old is None or old.shape != x.shape):
            raise RuntimeError("gated update: `old` must have the shape of the result")
        if base is not None and base.shape != x.shape:
            raise RuntimeError("update_multi: base and x differ in shape")
        out = torch.empty_like(x)
        N = x.shape[0]
        d.base, d.x, d.out, d.old = _ptr(base), _ptr(x), _ptr(out), _ptr(old)
        d.N, d.M, d.kind, d.step = N, x.numel() // max(N, 1), int(kind), float(step)
        outs.append(out)
        keep.extend((x, base, old))
    _lib.check(lib.advchain_update_multi(ctypes.byref(descs), len(items), _ptr(g), _stream()), "update_multi")
    return outs


class _Readback:
    """A few device floats copied to pinned host memory right where they are produced, with an event recorded behind
    the copy.  `values()` waits on THAT event only: by the time the backward asks, the copy finished long ago, so the
    read does not drain the kernels queued since (a `.item()` would enqueue its copy behind all of them)."""

    def __init__(self, dev_tensor):
        self.host = torch.empty(dev_tensor.shape, dtype=dev_tensor.dtype, pin_memory=True)
        self.host.copy_(dev_tensor, non_blocking=True)
        self.event = torch.cuda.Event()
        self.event.record(_stream_obj())
        self._vals = None

    def values(self):
        if self._vals is None:
            self.event.synchronize()
            self._vals = self.host.tolist()
        return self._vals



# ------------------------------------------------------------------------------------------------
# launch plans: the kernel selection of one solver call, recorded and then frozen (hipGraph replay)
# ------------------------------------------------------------------------------------------------
# The backward kernels are chosen from displacement bounds that the forward MEASURES and the host reads back (squaring_halo,
# warp_halo); the 3D chain reads a whole-batch norm for the reference's step rule.  A captured hipGraph cannot read anything
# back: a LaunchPlan records what those sites read during ordinary (eager) calls, freezes the values with a margin, and the
# capture takes every selection from the frozen plan.  Each site then enqueues advchain_bounds_check on what the replay
# measures against the interval its frozen selection is exact for; a raised flag tells the caller (the solver) to run
# that call again the ordinary way.  Results of a replay whose flag stays down are those of the ordinary path with the
# same selection (every formulation is tested against every other; the selection never changes values beyond that).
_PLAN = None


class PlanMismatch(RuntimeError):
    """The launch sequence met during a capture is not the recorded one."""


class _FrozenBounds(object):
    """Stands in for a _Readback when the plan is frozen: the bounds are host numbers known before the launch."""

    class _Done(object):
        @staticmethod
        def query():
            return True

        @staticmethod
        def synchronize():
            return None

    event = _Done

    def __init__(self, vals):
        self._vals = [float(v) for v in vals]

    def values(self):
        return self._vals


def _halo_threshold(h):
    """The displacement below which the bound `h` (squaring_halo / warp_halo) is exact; inf for a hint."""
    if h == -1:
        return 0.999
    if h < 0:
        return -h - 0.001
    return float("inf")


def _warp_halo_of(est, d):
    if not est == est:
        return 0
    if d == 3:
        return -1 if est < 0.999 else _halo_3d(est)
    return _halo_2d(est)


class LaunchPlan(object):
    """recording: sites append what they read (note); end_record() turns the read-backs into numbers and merges them with
    the calls recorded before (element-wise max).  freeze(): numbers x margin -> per-site selections + device intervals.
    frozen: take() hands the sites out in call order, check() enqueues the premise check."""

    def __init__(self, margin=1.3):
        # margin: the recorded maxima x margin pick the frozen selection.  Displacements vary with the random initial
        # parameters (cfg-2, second ascent step: 0.886 px over two recorded calls, 1.107 px in the third), and the levels of a
        # chain double, so every level sits at the same relative distance from ITS power-of-two threshold: a margin that is
        # too small is violated on all of them at once.  1.1 was (one call in five re-run the ordinary way); what a wide
        # margin costs is one formulation up on the levels within its reach of a threshold (+0.1 ms per cfg-2 call at 1.25)
        self.margin = float(margin)
        self.recording = True
        self.pending = []        # this call's sites while recording: (kind, payload)
        self.recorded = None     # merged: list of (kind, dict of numbers)
        self.calls = 0
        self.unstable = False    # two recorded calls did not visit the same sites
        self.frozen = None
        self.cursor = 0
        self.flag = None
        self.persistent = {}
        self.wanted_zeros = set()  # (tag, shape, device) of the persistent accumulators the recorded calls asked for
        self.violated = []       # diagnostics: which frozen bound a violated replay exceeded (filled by the recorded re-run)

    @property
    def is_frozen(self):
        return self.frozen is not None and not self.recording

    # -- recording
    def begin_record(self):
        """An ordinary call is about to be recorded.  A frozen selection stays in place (a captured graph keeps replaying it
        after this call); whatever an interrupted call left pending is dropped."""
        self.recording = True
        self.pending = []

    def thaw(self):
        """Drop the frozen selection (a new capture will freeze the merged record again)."""
        self.frozen = None
        self.persistent = {}

    def note(self, kind, **payload):
        self.pending.append((kind, payload))

    def end_record(self):
        sites = []
        for kind, pl in self.pending:
            if kind == "chain":
                sites.append((kind, {"n": pl["n"], "d": pl["d"], "vals": [float(v) for v in pl["rb"].values()]}))
            elif kind == "nsteps":
                sites.append((kind, {"n": pl["n"], "n_base": pl["n_base"]}))
            elif kind == "warp":
                sites.append((kind, {"d": pl["d"], "vals": [float(pl["rb"].values()[pl["idx"]])]}))
        self.pending = []
        # a recorded call that follows a violated replay measured what that replay measured: say which bound gave way
        last = self.frozen
        if last is not None and len(last) == len(sites):
            for i, (fz, (kind, rec)) in enumerate(zip(last, sites)):
                if "vals" in rec and fz["kind"] == kind:
                    his = fz["hi"].tolist()
                    for j, v in enumerate(rec["vals"]):
                        if j < len(his) and not (v < his[j]):
                            self.violated.append({"site": i, "kind": kind, "row": j, "measured": round(v, 4), "bound": round(his[j], 4),
                                                  "frozen_from": round(fz["bounds"].values()[j], 4)})
            del self.violated[:-64]
        if self.recorded is None:
            self.recorded = sites
        elif [(k, v.get("n"), v.get("d")) for k, v in sites] != [(k, v.get("n"), v.get("d")) for k, v in self.recorded]:
            self.unstable = True          # (another step count, another number of sites): the latest call wins
            self.recorded = sites
        else:
            for (_, old), (_, new) in zip(self.recorded, sites):
                if "vals" in old:
                    old["vals"] = [b if (b != b or b > a) else a for a, b in zip(old["vals"], new["vals"])]
        self.calls += 1
        self.recording = False

    # -- freezing
    def freeze(self, device):
        import numpy as np
        los, his, sites = [], [], []
        for kind, rec in self.recorded:
            if kind == "chain":
                n, d = rec["n"], rec["d"]
                vals = [v * self.margin if v == v else v for v in rec["vals"]]
                thr = [_halo_threshold(squaring_halo(vals[m], d)) for m in range(n)]
                thr.append(_halo_threshold(_warp_halo_of(vals[n], d)))
                thr += [float("inf")] * (len(vals) - n - 1)
                site = {"kind": kind, "n": n, "d": d, "bounds": _FrozenBounds(vals), "off": len(los), "len": len(thr)}
                los += [-float("inf")] * len(thr)
                his += thr
            elif kind == "nsteps":
                n, nb = rec["n"], rec["n_base"]
                # the rule of adv_morph.py:159-162 on sqrt(sum u^2): n is the smallest count >= n_base with norm / 2^n <= 0.5
                # (norm in (2^(n-2), 2^(n-1)]; as the half-open interval [lo, hi) of fp32 numbers the check kernel takes)
                up = lambda x: float(np.nextafter(np.float32(x), np.float32(np.inf)))
                hi = up(2.0 ** (n - 1))
                lo = up(2.0 ** (n - 2)) if n > nb else -float("inf")
                site = {"kind": kind, "n": n, "n_base": nb, "off": len(los), "len": 1}
                los.append(lo)
                his.append(hi)
            else:
                d = rec["d"]
                est = rec["vals"][0] * self.margin
                site = {"kind": kind, "d": d, "bounds": _FrozenBounds([est]), "off": len(los), "len": 1}
                los.append(-float("inf"))
                his.append(_halo_threshold(_warp_halo_of(est, d)))
            sites.append(site)
        f32 = torch.float32
        lo_t = torch.tensor(los or [0.0], dtype=f32).clamp(-3.0e38, 3.0e38).to(device)
        hi_t = torch.tensor(his or [0.0], dtype=f32).to(device)      # (inf stays inf: x < inf holds for every finite x)
        for sdict in sites:
            sdict["lo"] = lo_t[sdict["off"]:sdict["off"] + sdict["len"]]
            sdict["hi"] = hi_t[sdict["off"]:sdict["off"] + sdict["len"]]
        self.flag = torch.zeros(1, device=device, dtype=torch.int32)
        # what a replay measures, site after site, in ONE buffer (the producers write straight into their site's slice): one
        # check launch at the end of the replay instead of one per site (finish())
        self.measured = torch.zeros(max(1, len(los)), device=device, dtype=f32)
        self._lo_t, self._hi_t, self._deferred = lo_t, hi_t, 0
        self.frozen = sites
        self.recording = False
        self.cursor = 0
        self.persistent = {k: torch.zeros(k[1], device=device, dtype=f32) for k in sorted(self.wanted_zeros, key=repr) if k[2] == str(device)}

    # -- frozen
    def rewind(self):
        """Before the frozen sites are walked again (a capture; a launch-by-launch run of the frozen selection): the first
        site again, the measurement slots zeroed (the warp sites accumulate a maximum into theirs)."""
        self.cursor = 0
        self._deferred = 0
        self.measured.zero_()

    def slot(self, site):
        """Where the producer of this site's measurement writes it (a view of `measured`)."""
        return self.measured[site["off"]:site["off"] + site["len"]]

    def finish(self):
        """The premise check of everything measured into the slots since rewind(): one launch."""
        if self._deferred:
            n = self.measured.numel()
            _lib.check(_lib.load().advchain_bounds_check(_ptr(self.measured), _ptr(self._lo_t), _ptr(self._hi_t), n,
                                                         ctypes.c_void_p(self.flag.data_ptr()), _stream()), "bounds_check")
            self._deferred = 0

    def take(self, kind, **expect):
        if self.cursor >= len(self.frozen):
            raise PlanMismatch("launch plan: more %s sites than recorded" % kind)
        site = self.frozen[self.cursor]
        self.cursor += 1
        if site["kind"] != kind or any(site.get(k) != v for k, v in expect.items()):
            raise PlanMismatch("launch plan: met a %s site %r where a %s site was recorded" % (kind, expect, site["kind"]))
        return site

    def check(self, values, site):
        if values.numel() != site["len"]:
            raise PlanMismatch("launch plan: %d measured values for a site of %d" % (values.numel(), site["len"]))
        if values.data_ptr() == self.measured.data_ptr() + 4 * site["off"]:      # written into its slot: checked by finish()
            self._deferred += 1
            return
        _lib.check(_lib.load().advchain_bounds_check(_ptr(values), _ptr(site["lo"]), _ptr(site["hi"]), site["len"],
                                                     ctypes.c_void_p(self.flag.data_ptr()), _stream()), "bounds_check")


def grid_displacement(grid):
    """Max displacement (voxels) of a sampling grid, measured once per grid tensor (the entry -- a 1-float device tensor
    and, after the first read-back, its host value -- rides on the tensor object: the same deformation warps the image
    and then the prediction)."""
    hit = getattr(grid, "_advchain_disp", None)
    if hit is None or hit[2] != grid._version:
        plan = _PLAN
        key = (str(grid.device),) + tuple(grid.shape[2:])
        if plan is not None and plan.is_frozen:        # frozen plan: the recorded bound, and the premise check on the measured one
            site = plan.take("warp", d=grid.dim() - 2)
            plan.check(raw_max_displacement(grid.detach(), out=plan.slot(site)), site)
            hit = [site["bounds"], None, grid._version, 0, None]
        else:
            hit = [_Readback(raw_max_displacement(grid.detach())), None, grid._version, 0, key]
            if plan is not None:
                plan.note("warp", rb=hit[0], idx=0, d=grid.dim() - 2)
        grid._advchain_disp = hit
    return hit


def forward_hint(grid):
    """Displacement estimate (voxels) for the FORWARD warp by `grid`, without waiting for anything: the bound riding on
    the grid if its read-back has already arrived, else the last bound read for a grid of this shape (the deformation of
    the previous ascent step), else None.  Picks the forward kernel only."""
    hit = getattr(grid, "_advchain_disp", None)
    if hit is not None and hit[2] == grid._version:
        if hit[1] is None and hit[0].event.query():
            hit[1] = float(hit[0].values()[hit[3]])
            if len(hit) > 4 and hit[4] is not None:
                _WARP_HINTS[hit[4]] = hit[1]
        if hit[1] is not None:
            return hit[1]
    return _WARP_HINTS.get((str(grid.device),) + tuple(grid.shape[2:]))


def warp_halo(entry, d):
    """Displacement bound for the backward of a warp from the measured value (one 4-byte read-back per grid);
    negative = exact (see squaring_halo)."""
    if entry[1] is None:
        entry[1] = float(entry[0].values()[entry[3]])
    est = entry[1]
    if len(entry) > 4 and entry[4] is not None:
        _WARP_HINTS[entry[4]] = est
    return _warp_halo_of(est, d)


class _GridSample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inp, grid, interp, padding, clamp_grid, disp, hint=None, ride=None, ride_nonzero=False):
        inp, grid = _dev(inp, "input"), _dev(grid, "grid")
        ctx.save_for_backward(inp, grid)
        ctx.cfg = (interp, padding, clamp_grid)
        ctx.disp = disp
        if ride is None:
            return raw_grid_sample_fwd(inp, grid, interp, padding, clamp_grid, hint)
        # a one-channel rider (the solver's validity mask) through the same grid: second output, no gradient
        out, rout = raw_grid_sample_fwd_ride(inp, grid, _dev(ride, "rider"), interp, padding, clamp_grid, ride_nonzero, hint)
        ctx.mark_non_differentiable(rout)
        ctx.set_materialize_grads(False)     # (or the engine fills a zero gradient for the rider on every backward)
        return out, rout

    @staticmethod
    def backward(ctx, gout, _grider=None):
        inp, grid = ctx.saved_tensors
        interp, padding, clamp_grid = ctx.cfg
        need_in, need_grid = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if gout is None or not (need_in or need_grid):
            return (None,) * 9
        halo = warp_halo(ctx.disp, inp.dim() - 2) if (ctx.disp is not None and need_in) else 0
        gin, ggrid = raw_grid_sample_bwd(_dev(gout, "grad"), inp, grid, interp, padding, clamp_grid, need_in, need_grid,
                                         halo)
        return (gin, ggrid) + (None,) * 7


@_on_tensor_device
def grid_sample(inp, grid, interp="bilinear", padding_mode="zeros", clamp_grid=False, ride=None, ride_nonzero=False):
    """F.grid_sample(inp, grid^T, mode, padding_mode, align_corners=True) with a PLANAR grid (N,d,...).
    ride: a one-channel tensor (N,1,...) warped through the same grid in the same call (linear / nearest only) -> returns
    (out, ride_out); ride_out carries no gradient; ride_nonzero: ride_out = (warp(ride) != 0) as 0 / 1."""
    code = interp_code(interp)
    nd = inp.dim() - 2
    if nd not in (2, 3) or grid.dim() != inp.dim() or grid.shape[1] != nd:
        raise RuntimeError("grid_sample(): expected a planar grid of shape (N, %d, ...) for a %d-D input, got %s"
                           % (nd, inp.dim(), tuple(grid.shape)))
    if grid.shape[0] != inp.shape[0]:
        raise RuntimeError("grid_sample(): expected grid and input to have same batch size, but got input with sizes "
                           "%s and grid with sizes %s" % (list(inp.shape), list(grid.shape)))
    _same_device(inp, grid)
    if code == 2:
        if nd != 2:      # F.grid_sample's own message for 5-D input
            raise RuntimeError("grid_sampler(): bicubic interpolation only supports 4D input")
        if ride is not None:
            raise NotImplementedError("grid_sample(): no rider with bicubic interpolation")
        return _GridSampleBicubic.apply(inp, torch.clamp(grid, -1, 1) if clamp_grid else grid, pad_code(padding_mode))
    disp = None
    if (ADAPTIVE_HALO and code == 0 and torch.is_grad_enabled() and inp.requires_grad and grid.is_cuda
            and inp.shape[2:] == grid.shape[2:] and grid.dtype == torch.float32 and grid.is_contiguous()):
        disp = grid_displacement(grid)       # the backward sizes its halo / picks the gather form from it
    hint = forward_hint(grid) if (code == 0 and nd == 3 and grid.is_cuda) else None
    if ride is not None:
        if tuple(ride.shape) != (inp.shape[0], 1) + tuple(inp.shape[2:]):
            raise RuntimeError("grid_sample(): the rider must be (N, 1, ...) of the input's size, got %s" % (tuple(ride.shape),))
        _same_device(inp, ride)
        return _GridSample.apply(inp, grid, code, pad_code(padding_mode), bool(clamp_grid), disp, hint, ride.detach(),
                                 bool(ride_nonzero))
    return _GridSample.apply(inp, grid, code, pad_code(padding_mode), bool(clamp_grid), disp, hint)


class _GridSampleBicubic(torch.autograd.Function):
    """F.grid_sample(inp, grid^T, mode='bicubic', padding_mode, align_corners=True) with a PLANAR grid (N,2,OH,OW); 2D only
    (adv_morph.py:255-258,546-557 with forward_interp / backward_interp = 'bicubic')."""

    @staticmethod
    def forward(ctx, inp, grid, padding):
        inp, grid = _dev(inp, "input"), _dev(grid, "grid")
        N, C = inp.shape[:2]
        out = torch.empty((N, C) + tuple(grid.shape[2:]), device=inp.device, dtype=torch.float32)
        _lib.check(_lib.load().advchain_grid_sample_bicubic2d_fwd(_ptr(inp), _ptr(grid), _ptr(out), N, C,
                                                                  _lib.dims_array(inp.shape[2:]), _lib.dims_array(grid.shape[2:]),
                                                                  padding, _stream()), "grid_sample_bicubic2d_fwd")
        ctx.save_for_backward(inp, grid)
        ctx.padding = padding
        return out

    @staticmethod
    def backward(ctx, gout):
        inp, grid = ctx.saved_tensors
        need_in, need_grid = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if not (need_in or need_grid):
            return None, None, None
        gin = torch.empty_like(inp) if need_in else None
        ggrid = torch.empty_like(grid) if need_grid else None
        N, C = inp.shape[:2]
        _lib.check(_lib.load().advchain_grid_sample_bicubic2d_bwd(_ptr(_dev(gout, "grad")), _ptr(inp), _ptr(grid), _ptr(gin), _ptr(ggrid),
                                                                  N, C, _lib.dims_array(inp.shape[2:]), _lib.dims_array(grid.shape[2:]),
                                                                  ctx.padding, _stream()), "grid_sample_bicubic2d_bwd")
        return gin, ggrid, None


class _AffineGrid2D(torch.autograd.Function):
    """F.affine_grid(theta, (N, C, H, W), align_corners=True) as a planar grid (N,2,H,W) (bicubic affine warps only: the
    linear / nearest ones never materialise the grid)."""

    @staticmethod
    def forward(ctx, theta, H, W):
        theta = _dev(theta, "theta")
        N = theta.shape[0]
        grid = torch.empty((N, 2, H, W), device=theta.device, dtype=torch.float32)
        _lib.check(_lib.load().advchain_affine_grid2d_fwd(_ptr(theta), _ptr(grid), N, _lib.dims_array((H, W)), _stream()),
                   "affine_grid2d_fwd")
        ctx.dims = (H, W)
        return grid

    @staticmethod
    def backward(ctx, ggrid):
        H, W = ctx.dims
        ggrid = _dev(ggrid, "grad")
        N = ggrid.shape[0]
        lib = _lib.load()
        dims = _lib.dims_array((H, W))
        ws = torch.empty(max(1, lib.advchain_affine_grid2d_bwd_workspace(N, dims)), device=ggrid.device, dtype=torch.float32)
        gth = torch.empty((N, 2, 3), device=ggrid.device, dtype=torch.float32)
        _lib.check(lib.advchain_affine_grid2d_bwd(_ptr(ggrid), _ptr(gth), _ptr(ws), N, dims, _stream()), "affine_grid2d_bwd")
        return gth, None, None


class _AffineWarp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inp, theta, interp, padding, ride=None, ride_nonzero=False):
        inp, theta = _dev(inp, "input"), _dev(theta, "theta")
        N, C = inp.shape[:2]
        nd = inp.dim() - 2
        out = torch.empty_like(inp)
        ctx.save_for_backward(inp, theta)
        ctx.cfg = (interp, padding)
        if ride is not None:     # a one-channel rider (the solver's validity mask) under the same theta: no gradient
            ride = _dev(ride, "rider")
            rout = torch.empty_like(ride)
            _lib.check(_lib.load().advchain_affine_warp_fwd_ride(_ptr(inp), _ptr(theta), _ptr(out), _ptr(ride), _ptr(rout), N, C,
                                                                 nd, _lib.dims_array(inp.shape[2:]), interp, padding,
                                                                 int(bool(ride_nonzero)), _stream()), "affine_warp_fwd_ride")
            ctx.mark_non_differentiable(rout)
            ctx.set_materialize_grads(False)     # (or the engine fills a zero gradient for the rider on every backward)
            return out, rout
        _lib.check(_lib.load().advchain_affine_warp_fwd(_ptr(inp), _ptr(theta), _ptr(out), N, C, nd,
                                                        _lib.dims_array(inp.shape[2:]), interp, padding, _stream()),
                   "affine_warp_fwd")
        return out

    @staticmethod
    def backward(ctx, gout, _grider=None):
        inp, theta = ctx.saved_tensors
        interp, padding = ctx.cfg
        need_in, need_th = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if gout is None or not (need_in or need_th):
            return (None,) * 6
        lib = _lib.load()
        N, C = inp.shape[:2]
        nd = inp.dim() - 2
        dims = _lib.dims_array(inp.shape[2:])
        gin = torch.empty_like(inp) if need_in else None
        gth = torch.empty_like(theta) if need_th else None
        ws = torch.empty(max(1, lib.advchain_affine_warp_bwd_workspace(N, nd, dims)), device=inp.device,
                         dtype=torch.float32)
        _lib.check(lib.advchain_affine_warp_bwd(_ptr(_dev(gout, "grad")), _ptr(inp), _ptr(theta), _ptr(gin), _ptr(gth),
                                                _ptr(ws), N, C, nd, dims, interp, padding, _stream()), "affine_warp_bwd")
        return gin, gth, None, None, None, None


@_on_tensor_device
def affine_warp(inp, theta, interp="bilinear", padding_mode="zeros", ride=None, ride_nonzero=False):
    """F.grid_sample(inp, F.affine_grid(theta, inp.size(), align_corners=True), ..., align_corners=True).
    ride / ride_nonzero: as for grid_sample -> returns (out, ride_out)."""
    nd = inp.dim() - 2
    if tuple(theta.shape) != (inp.shape[0], nd, nd + 1):
        raise RuntimeError("Expected a batch of %dD affine matrices of shape Nx%dx%d for size %s. Got %s."
                           % (nd, nd, nd + 1, list(inp.shape), list(theta.shape)))
    _same_device(inp, theta)
    code = interp_code(interp)
    if code == 2:
        if nd != 2:
            raise RuntimeError("grid_sampler(): bicubic interpolation only supports 4D input")
        if ride is not None:
            raise NotImplementedError("affine_warp(): no rider with bicubic interpolation")
        grid = _AffineGrid2D.apply(theta, int(inp.shape[2]), int(inp.shape[3]))
        return _GridSampleBicubic.apply(inp, grid, pad_code(padding_mode))
    if ride is not None:
        if tuple(ride.shape) != (inp.shape[0], 1) + tuple(inp.shape[2:]):
            raise RuntimeError("affine_warp(): the rider must be (N, 1, ...) of the input's size, got %s" % (tuple(ride.shape),))
        _same_device(inp, ride)
        return _AffineWarp.apply(inp, theta, code, pad_code(padding_mode), ride.detach(), bool(ride_nonzero))
    return _AffineWarp.apply(inp, theta, code, pad_code(padding_mode))


class _AffineTheta(torch.autograd.Function):
    @staticmethod
    def forward(ctx, param, cfg, param_scale, nd):
        param = _dev(param, "param")
        N = param.shape[0]
        theta = torch.empty((N, nd, nd + 1), device=param.device, dtype=torch.float32)
        theta_inv = torch.empty_like(theta)
        cfg_arr = _lib.float_array(cfg)
        _lib.check(_lib.load().advchain_affine_theta_fwd(_ptr(param), cfg_arr, float(param_scale), _ptr(theta),
                                                         _ptr(theta_inv), N, nd, _stream()), "affine_theta_fwd")
        ctx.save_for_backward(param)
        ctx.cfg = (cfg_arr, float(param_scale), nd)
        return theta, theta_inv

    @staticmethod
    def backward(ctx, gtheta, gtheta_inv):
        (param,) = ctx.saved_tensors
        cfg_arr, scale, nd = ctx.cfg
        gparam = torch.empty_like(param)
        gtheta = None if gtheta is None else _dev(gtheta, "grad_theta")
        gtheta_inv = None if gtheta_inv is None else _dev(gtheta_inv, "grad_theta_inv")
        _lib.check(_lib.load().advchain_affine_theta_bwd(_ptr(param), cfg_arr, scale, _ptr(gtheta), _ptr(gtheta_inv),
                                                         _ptr(gparam), param.shape[0], nd, _stream()),
                   "affine_theta_bwd")
        return gparam, None, None, None


@_on_tensor_device
def affine_theta(param, cfg, param_scale, nd):
    """(N,5|9) bounded parameters -> (theta, theta^-1), each (N, nd, nd+1)."""
    return _AffineTheta.apply(param, tuple(float(c) for c in cfg), float(param_scale), int(nd))


class _Axpy(torch.autograd.Function):
    """out = x + a * y  (AdvNoise.forward, adv_noise.py:81-84)."""

    @staticmethod
    def forward(ctx, x, y, a):
        ctx.a = a
        return raw_axpy(_dev(x, "x"), _dev(y, "y"), a)

    @staticmethod
    def backward(ctx, g):
        g = _dev(g, "grad")
        gx = g if ctx.needs_input_grad[0] else None
        # (a == 1: 1.0 * g is g, bit for bit -- no launch; the notebook's epsilon for AdvNoise)
        gy = (g if ctx.a == 1.0 else raw_axpy(None, g, ctx.a)) if ctx.needs_input_grad[1] else None
        return gx, gy, None


@_on_tensor_device
def axpy(x, y, a):
    """x + a * y with torch's broadcasting rules (the reference writes ``data + epsilon * param``)."""
    if x.shape != y.shape:
        x, y = torch.broadcast_tensors(x, y)      # raises like torch when the shapes do not broadcast
    _same_device(x, y)
    return _Axpy.apply(x, y, float(a))


class _BiasApply(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cp, data, tables, eps, use_log, cp_scale):
        cp, data = _dev(cp, "control points"), _dev(data, "data")
        N, C = data.shape[:2]
        out = torch.empty_like(data)
        field = torch.empty((N, 1) + tuple(data.shape[2:]), device=data.device, dtype=torch.float32)
        _lib.check(_lib.load().advchain_bias_field_fwd(_ptr(cp), _ptr(data), _ptr(out), _ptr(field), _ptr(tables.itab),
                                                       _ptr(tables.ftab), _lib.dims_array(tables.S),
                                                       _lib.dims_array(tables.g), _lib.dims_array(tables.B), N, C,
                                                       float(eps), int(use_log), float(cp_scale), _stream()),
                   "bias_field_fwd")
        ctx.save_for_backward(cp, data)
        ctx.cfg = (tables, float(eps), int(use_log), float(cp_scale))
        ctx.mark_non_differentiable(field)
        ctx.set_materialize_grads(False)      # (the engine would otherwise zero-fill a full-size gradient for `field`)
        return out, field

    @staticmethod
    def backward(ctx, gout, _gfield):
        cp, data = ctx.saved_tensors
        tables, eps, use_log, cp_scale = ctx.cfg
        need_cp, need_data = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if gout is None or not (need_cp or need_data):
            return None, None, None, None, None, None
        N, C = data.shape[:2]
        gL = torch.empty((N, 1) + tuple(data.shape[2:]), device=data.device, dtype=torch.float32) if need_cp else None
        gdata = torch.empty_like(data) if need_data else None
        _lib.check(_lib.load().advchain_bias_field_bwd(_ptr(cp), _ptr(data), _ptr(_dev(gout, "grad")), _ptr(gL),
                                                       _ptr(gdata), _ptr(tables.itab), _ptr(tables.ftab),
                                                       _lib.dims_array(tables.S), _lib.dims_array(tables.g),
                                                       _lib.dims_array(tables.B), N, C, eps, use_log, cp_scale,
                                                       _stream()), "bias_field_bwd")
        gcp = raw_tp_adjoint(gL, tables).reshape(cp.shape) if need_cp else None
        return gcp, gdata, None, None, None, None


@_on_tensor_device
def bias_apply(cp, data, tables, eps, use_log=True, cp_scale=1.0):
    """(data * clipped_bias_field(cp), clipped_bias_field)."""
    if cp.shape[0] != data.shape[0] or tuple(data.shape[2:]) != tuple(tables.full_dims) \
            or tuple(cp.shape[2:]) != tuple(int(g) for g in tables.g[3 - tables.ndim:]):
        raise RuntimeError("bias field: control points %s / data %s do not match the configured lattice %s -> image %s"
                           % (list(cp.shape), list(data.shape), list(tables.g[3 - tables.ndim:]),
                              list(tables.full_dims)))
    _same_device(cp, data)
    return _BiasApply.apply(cp, data, tables, eps, use_log, cp_scale)


@_on_tensor_device
def bias_field_only(cp, tables, eps, use_log=True, cp_scale=1.0):
    cp = _dev(cp.detach(), "control points")
    N = cp.shape[0]
    field = torch.empty((N, 1) + tuple(tables.full_dims), device=cp.device, dtype=torch.float32)
    _lib.check(_lib.load().advchain_bias_field_fwd(_ptr(cp), None, None, _ptr(field), _ptr(tables.itab),
                                                   _ptr(tables.ftab), _lib.dims_array(tables.S),
                                                   _lib.dims_array(tables.g), _lib.dims_array(tables.B), N, 1,
                                                   float(eps), int(use_log), float(cp_scale), _stream()),
               "bias_field_fwd")
    return field


class _DemonsField(torch.autograd.Function):
    """Low-res velocity -> un-clamped sampling grid (DemonsCompose, adv_morph.py:454-491).

    forward:  gauss(scale*v) -> linear upsample -> phi0 = id + u/2^n -> n x (phi <- phi o phi)
              -> pos = (phi_n - phi0) + id -> gauss(border_identity(pos) - id) + id
    The final clamp(-1,1) (adv_morph.py:490, 304-305) is applied by the sampler on load.
    backward: the hand-written adjoint of the same chain (saved: phi_0..phi_{n-1}, pos).

    opts = (num_steps, smooth_iter, sigma, positions_only[, taps]) -- the attributes of AdvMorph the reference reads in
    DemonsCompose (adv_morph.py:236-242,461-471): defaults (8, 1, 1.0, False); taps = the Gaussian window length when it
    is not the rule's 2 * int(4 sigma + 0.5) + 1 (a gaussian_ks above it, adv_morph.py:393-398).  positions_only returns `pos` itself (the
    caller composes it with an initial deformation other than the identity and / or skips the final smoothing)."""

    @staticmethod
    def forward(ctx, vel, scale, tables, nsteps_rule, reduce_sumsq, pair=False, opts=None):
        vel = _dev(vel, "velocity")
        ctx.pair = bool(pair)
        n_base, smooth_iter, sigma, pos_only = (tuple(opts) + (None,))[:4] if opts is not None else (8, 1, 1.0, False)
        w9 = gauss9(sigma, opts[4] if (opts is not None and len(opts) > 4) else None)
        if len(w9) != 9 and not pos_only:
            raise NotImplementedError("the fused final smoothing exists for the 9-tap window only: ask for the positions")
        ctx.opts = (int(smooth_iter), w9, bool(pos_only))
        s1 = None
        # 2D paired fields with the default options: ONE C call enqueues the whole direction (advchain_demons_compose_pair_fwd:
        # the launches of the separate calls below, in their order)
        composite = (COMPOSITE and pair and vel.shape[1] == 2 and not nsteps_rule and int(smooth_iter) == 1 and not pos_only
                     and len(w9) == 9 and vel.is_contiguous() and vel[0, 0].numel() <= 4096
                     and getattr(tables, "dense_inner", None) is not None)
        ctx.composite = composite
        if pair and not composite:      # the batch [v; -v]: both fields of a solver step from one chain; returns (field(+v), field(-v))
            s1 = raw_gauss_small_pair(vel, scale, weights=w9)         # the negated copy is never materialised
            if s1 is None:
                vel = torch.cat([vel, -vel], 0)
        d = vel.shape[1]
        if composite:
            s1 = torch.empty((2 * vel.shape[0],) + tuple(vel.shape[1:]), device=vel.device, dtype=torch.float32)
        elif s1 is None:
            s1 = raw_gauss(vel, d, pre=1, scale=scale, weights=w9)
        for _ in range(int(smooth_iter) - 1):      # smooth_iter > 1 (adv_morph.py:386-387): the same window again
            s1 = raw_gauss(s1, d, weights=w9)
        N = s1.shape[0]
        n = int(n_base)
        plan = _PLAN
        frozen = plan is not None and plan.is_frozen
        norm_rb = None
        if nsteps_rule:  # 3D: whole-batch Frobenius norm of u / 2^n must not exceed 0.5 (adv_morph.py:159-162)
            slots = torch.zeros(64, device=vel.device, dtype=torch.float32)
            # pair: the batch is [v; -v] -- the rule is the reference's, over ONE field's batch (both halves agree)
            raw_tp_interp(s1[:N // 2] if pair else s1, tables, d, want_out=False, sumsq=slots)
            ss = slots.sum().reshape(1)
            if reduce_sumsq is not None:
                ss = reduce_sumsq(ss)
            if frozen:       # the recorded count; the replay checks on the device that the rule still gives it
                site = plan.take("nsteps", n_base=n)
                n = site["n"]
                plan.check(torch.sqrt(ss, out=plan.slot(site)), site)
            else:
                # no `.item()` (round 5): the norm travels to the host behind an event of its own while the chain is
                # enqueued with the count the previous field of this shape needed; the count is verified below, once the
                # chain is queued -- the wait is for a kernel at the HEAD of the queue, so it drains nothing -- and only a
                # count that changed (the first call of a shape, an ascent that crosses a power of two) enqueues the chain
                # a second time.  The results are those of the reference's rule either way.
                nkey = (str(s1.device), tuple(s1.shape), int(n_base), HINT_SLOT)
                norm_rb = _Readback(ss.sqrt())
                n = max(n, _NSTEPS_HINT.get(nkey, n))
        mark = None if plan is None else len(plan.pending)
        while True:
            out = _DemonsField._enqueue_chain(ctx, vel, s1, tables, d, n, scale, w9, pos_only, composite, plan, frozen, pair)
            if norm_rb is None:
                break
            norm, want = float(norm_rb.values()[0]), int(n_base)
            while norm / (2.0 ** want) > 0.5:
                want += 1
            norm_rb = None
            _NSTEPS_HINT[nkey] = want
            NSTEPS_STATS["chains"] += 1
            if plan is not None:         # (a frozen plan meets the count first, then the chain: keep that order in the record)
                if want != n:
                    del plan.pending[mark:]
                plan.pending.insert(mark, ("nsteps", dict(n=want, n_base=int(n_base))))
            if want == n:
                break
            NSTEPS_STATS["respeculated"] += 1
            n = want
        return out

    @staticmethod
    def _enqueue_chain(ctx, vel, s1, tables, d, n, scale, w9, pos_only, composite, plan, frozen, pair):
        N = s1.shape[0]
        inv = 1.0 / (2.0 ** n)
        # row m of `disp`: max-slots for the displacement of phis[m], written by the kernel that produces it; row n: the
        # sampling positions `pos`, which bound the returned grid (clipping to [-1,1] and the normalised Gaussian only
        # shrink a displacement).  Read back once (asynchronously): the backward sizes every step exactly from it.
        # (row n + 1, slot 0: the flag of the fused 2D squarings -- 1 when a window moved too far for them and the ordinary
        # launches ran instead; zeroed with the rest by the row-maxima kernel below)
        disp = _persistent_zeros("disp", (n + 2, DISP_SLOTS), vel.device) if ADAPTIVE_HALO else None
        row = (lambda m: None) if disp is None else (lambda m: disp[m])
        # what the squarings of the PREVIOUS field of this shape measured (a field changes little between two ascent
        # steps): picks the forward kernel per squaring, nothing else
        # ... of the same POSITION in the caller's trajectory when it says so (the solver numbers its ascent steps: the field
        # of step i resembles step i of the previous call -- 0.14 / 0.20 / 0.24 / 0.32 px after steps 1..4 at cfg-2 against 0.03
        # for a fresh draw -- far better than it resembles step i - 1 of this call)
        key = (str(s1.device), tuple(s1.shape), n, HINT_SLOT)
        site = None
        if frozen:           # every selection of this chain from the plan (its own recorded displacements, with the margin)
            if disp is None:
                raise PlanMismatch("launch plan: a frozen chain needs the measured displacements (ADVCHAIN_NO_ADAPTIVE_HALO is set)")
            site = plan.take("chain", n=n, d=d)
            hints = site["bounds"].values()
        else:
            pend = _PENDING_BOUNDS.pop(key, None)      # a chain whose backward never ran (the final pass): its read-back, if it has arrived
            if pend is not None and pend.event.query():
                _note_chain_bounds(key, pend.values(), n)
            hints = _CHAIN_HINTS.get(key)
        full = (N, d) + tuple(tables.full_dims)
        phi0 = torch.empty(full, device=vel.device, dtype=torch.float32) if composite else \
            raw_tp_interp(s1, tables, d, add_identity=True, scale=inv, disp_out=row(0))
        # the n squarings: one C call (advchain_expo_chain_fwd), phi_1..phi_{n-1} in one stacked buffer
        fields = torch.empty((n - 1,) + tuple(phi0.shape), device=phi0.device, dtype=torch.float32)
        pos = torch.empty_like(phi0)
        harr = None if hints is None else (ctypes.c_int32 * n)(*[(_hint_bits(hints[m]) >> 8) | (_fine_bits(hints[m]) << 8)
                                                                 for m in range(n)])
        if COUNT_FUSED and harr is not None and disp is not None and FUSE_2D:      # (tests: which formulation the chain takes)
            FUSE_STATS["fused_levels"] += _lib.load().advchain_expo_chain_fused_levels(phi0.shape[0], d, _lib.dims_array(phi0.shape[2:]), n, harr)
        try:
            rows_max = rc = None
            if composite:
                q = torch.empty_like(phi0)
                rows_max = None if disp is None else (plan.slot(site) if site is not None else
                                                      torch.empty(disp.shape[0], device=vel.device, dtype=torch.float32))
                rc = _lib.load().advchain_demons_compose_pair_fwd(
                    _ptr(vel), _ptr(s1), _ptr(phi0), _ptr(fields), _ptr(pos), _ptr(q), _ptr(disp), _ptr(rows_max), _ptr(tables.itab),
                    _ptr(tables.ftab), _lib.dims_array(tables.S), _lib.dims_array(tables.g), _lib.dims_array(tables.B),
                    vel.shape[0], d, n, harr, w9, float(scale), float(inv), int(bool(FUSE_2D)), _stream())
                if rc == -2:        # (a launch would not take the shape; nothing was enqueued: the separate calls)
                    composite = ctx.composite = False
                    raw_gauss_small_pair_into(vel, s1, scale, w9)
                    raw_tp_interp(s1, tables, d, add_identity=True, scale=inv, disp_out=row(0), out=phi0)
                else:
                    _lib.check(rc, "demons_compose_pair_fwd")
            if not composite:
                _lib.check(_lib.load().advchain_expo_chain_fwd(_ptr(phi0), _ptr(fields), _ptr(pos), phi0.shape[0], d,
                                                               _lib.dims_array(phi0.shape[2:]), n, _ptr(disp), harr,
                                                               None if (disp is None or not FUSE_2D) else _ptr(disp[n + 1]), _stream()),
                           "expo_chain_fwd")
                q = pos if pos_only else raw_gauss(pos, d, pre=2, post=1, weights=w9)
                rows_max = None if disp is None else raw_slot_rows_max(disp, reset=True, out=None if site is None else plan.slot(site))
        except BaseException:
            # (the slots, the fused-chain flag and its barrier counter may hold partial state: the next chain starts from a
            # fresh zero-filled accumulator)
            if disp is not None:
                _forget_persistent(disp)
            raise
        ctx.save_for_backward(pos, phi0, fields)
        ctx.frozen = site is not None
        if site is not None:
            plan.check(rows_max, site)
            ctx.disp = site["bounds"]
        else:
            ctx.disp = None if rows_max is None else _Readback(rows_max)
            if ctx.disp is not None:
                _PENDING_BOUNDS[key] = ctx.disp
                if plan is not None:
                    plan.note("chain", rb=ctx.disp, n=n, d=d)
        global _LAST_FIELD_BOUND
        _LAST_FIELD_BOUND = None if ctx.disp is None else (ctx.disp, n)
        ctx.cfg = (scale, tables, inv, d)
        ctx.nsteps = n
        ctx.hint_key = key
        if pair:
            return q[:N // 2], q[N // 2:]
        return q

    @staticmethod
    def backward(ctx, *grads):
        pos, phi0, fields = ctx.saved_tensors
        scale, tables, inv, d = ctx.cfg
        # pair: one contiguous gradient for the batch [v; -v] (autograd's own route -- two slice_backward zero-fills of
        # the whole batch, two copies and an add per chain -- cost more than the concatenation)
        smooth_iter, w9, pos_only = ctx.opts
        gq = tuple(_dev(g, "grad") for g in grads) if ctx.pair else _dev(grads[0], "grad")
        composite = (ctx.composite and TILED_SCATTER and ctx.pair and gq[0].is_contiguous() and gq[1].is_contiguous()
                     and gq[0].shape == gq[1].shape)
        if composite:
            gpos = torch.empty_like(pos)
        elif pos_only:
            gpos = torch.cat(gq, 0) if ctx.pair else gq.contiguous()
        else:
            gpos = raw_gauss(gq, d, post=2, aux=pos, weights=w9)          # adjoint of gauss(border_identity(.) - id) + id
        g = gpos                                          # d/d phi_n
        ws = _scatter_workspace(gpos.shape[0], gpos.shape[2:], gpos.device) if TILED_SCATTER else None
        # squaring m composes a field whose displacement is ~2^(m-n) of the total: the early steps are sub-voxel and
        # take the gather-form adjoint.  The bound is a performance hint only (larger displacements stay correct
        # through the overflow list); it comes from the displacement measured in forward (one 4-byte read-back).
        n = ctx.nsteps
        if ctx.disp is not None:
            dm = ctx.disp.values()
            if not ctx.frozen:
                _PENDING_BOUNDS.pop(ctx.hint_key, None)
                _note_chain_bounds(ctx.hint_key, dm, n)
            halos = [squaring_halo(dm[m], d) for m in range(n - 1, -1, -1)]
        else:
            big = 2 if d == 3 else 16
            halos = [big, big, max(1, big // 2)] + [1 if d == 3 else 2] * n
        halos = halos[:n]
        if composite:      # ONE C call for the whole direction (advchain_demons_compose_pair_bwd: the launches of the calls below)
            N2 = gpos.shape[0]
            S, gg = list(tables.S), list(tables.g)
            g, scratch = torch.empty_like(gpos), torch.empty_like(gpos)
            t1 = torch.empty((N2, d, S[1], gg[2]), device=gpos.device, dtype=torch.float32)
            gs1 = torch.empty((N2, d, gg[1], gg[2]), device=gpos.device, dtype=torch.float32)
            gvel = torch.empty((N2 // 2, d, gg[1], gg[2]), device=gpos.device, dtype=torch.float32)
            wd, wlo, WB = tables.dense_inner
            rc = _lib.load().advchain_demons_compose_pair_bwd(
                _ptr(gq[0]), _ptr(gq[1]), _ptr(pos), _ptr(phi0), _ptr(fields), _ptr(gpos), _ptr(g), _ptr(scratch), _ptr(ws), _ptr(t1),
                _ptr(gs1), _ptr(gvel), (ctypes.c_int32 * n)(*[int(h) for h in halos]), _ptr(tables.itab), _ptr(tables.ftab), _ptr(wd),
                _ptr(wlo), int(WB), _lib.dims_array(S), _lib.dims_array(gg), _lib.dims_array(tables.B), N2 // 2, d, n, w9, float(scale),
                float(inv), _stream())
            if rc != -2:
                _lib.check(rc, "demons_compose_pair_bwd")
                return gvel, None, None, None, None, None, None
            gpos = raw_gauss(gq, d, post=2, aux=pos, weights=w9)       # (nothing was enqueued: the separate calls)
            g = gpos
        if ws is None:     # (global-atomic A/B path: per-step calls with zero-filled outputs)
            phis = [phi0] + list(fields.unbind(0))
            for i, phi in enumerate(reversed(phis)):
                g = raw_compose_self_bwd(g, phi, None, chain=False, halo=halos[i])
        else:
            g, scratch = torch.empty_like(gpos), torch.empty_like(gpos)
            _lib.check(_lib.load().advchain_expo_chain_bwd(_ptr(gpos), _ptr(phi0), _ptr(fields), _ptr(g), _ptr(scratch), _ptr(ws),
                                                           (ctypes.c_int32 * n)(*[int(h) for h in halos]), gpos.shape[0], d,
                                                           _lib.dims_array(gpos.shape[2:]), n, _stream()), "expo_chain_bwd")
        # phi0 also enters through '- phi0' (Q1 aliasing): total = g - gpos ; u = (phi0 - id) * 2^n
        gs1 = raw_tp_adjoint(g, tables, gfull2=gpos, scale=inv)
        for _ in range(smooth_iter - 1):           # (the zero-padded symmetric window is its own adjoint)
            gs1 = raw_gauss(gs1, d, weights=w9)
        gvel = raw_gauss_small_pair(gs1, scale, adjoint=True, weights=w9) if ctx.pair else None
        if gvel is None:
            gvel = raw_gauss(gs1, d, pre=1, scale=scale, weights=w9)
            if ctx.pair:
                h = gvel.shape[0] // 2
                gvel = gvel[:h] - gvel[h:]
        return gvel, None, None, None, None, None, None


_LAST_FIELD_BOUND = None
HINT_SLOT = 0      # set by the solver: which step of its loop the next chain belongs to (keys the kernel-selection hints)
FUSE_STATS = {"chains": 0, "refused": 0, "fused_levels": 0}     # chains whose backward read its bounds / whose fused forward squarings fell back / (COUNT_FUSED) squarings the forward chains ran fused
COUNT_FUSED = False   # tests: ask advchain_expo_chain_fused_levels what every forward chain takes (one host call per chain)


class _HintCache(dict):
    """A small bounded map (kernel-selection hints keyed by device and shape; cleared when it outgrows its cap)."""
    CAP = 256

    def __setitem__(self, key, value):
        if len(self) >= self.CAP and key not in self:
            self.clear()
        dict.__setitem__(self, key, value)


_NSTEPS_HINT = _HintCache()    # (device, smoothed velocity shape, n_base) -> the step count the 3D rule gave the last field of that shape
NSTEPS_STATS = {"chains": 0, "respeculated": 0}      # 3D chains enqueued with a guessed count / enqueued again with the right one
_PENDING_BOUNDS = _HintCache()  # chain key -> the read-back of the last forward of that key, until somebody looks at it


def _note_chain_bounds(key, dm, n):
    """The displacements phi_0..phi_n (+ the deficit of the fused 2D squarings) a chain measured become the kernel-selection
    hints of the next chain with the same key."""
    FUSE_STATS["chains"] += 1
    if len(dm) > n + 1 and dm[n + 1] > 0:      # some window could not do all the levels the hints promised
        FUSE_STATS["refused"] += 1
    _CHAIN_HINTS[key] = list(dm)


_CHAIN_HINTS = _HintCache()     # (device, velocity shape, n, slot) -> displacement of phi_0..phi_n measured by the last backward of such a chain
_WARP_HINTS = _HintCache()      # (device, spatial dims) -> displacement of the last grid of that shape whose bound was read


def _ride_key(rb, q):
    """Key of the global warp-hint table for a bound riding on grid `q` (None: a frozen plan's bound is not a measurement)."""
    return None if isinstance(rb, _FrozenBounds) else (str(q.device),) + tuple(q.shape[2:])


@_on_tensor_device
def demons_field(vel, scale, tables, nsteps_rule, reduce_sumsq=None, opts=None):
    global _LAST_FIELD_BOUND
    _LAST_FIELD_BOUND = None
    q = _DemonsField.apply(vel, float(scale), tables, bool(nsteps_rule), reduce_sumsq, False, opts)
    if opts is not None and opts[3]:       # positions only: no bound rides on them (the caller composes them further)
        _LAST_FIELD_BOUND = None
        return q
    if _LAST_FIELD_BOUND is not None:      # the displacement bound rides on the grid: no second measurement by the warps
        rb, idx = _LAST_FIELD_BOUND
        q._advchain_disp = [rb, None, q._version, idx, _ride_key(rb, q)]
        _LAST_FIELD_BOUND = None
    return q


class _GaussSmooth(torch.autograd.Function):
    """x -> G * x, the zero-padded separable 9-tap Gaussian of every (n, c) plane (depthwise conv of adv_morph.py:377-452);
    the symmetric window is its own adjoint."""

    @staticmethod
    def forward(ctx, x, sigma, taps=None):
        x = _dev(x, "field")
        ctx.w9 = gauss9(sigma, taps)
        return raw_gauss(x, x.shape[1], weights=ctx.w9)

    @staticmethod
    def backward(ctx, g):
        g = _dev(g, "grad")
        return raw_gauss(g, g.shape[1], weights=ctx.w9), None, None


@_on_tensor_device
def gauss_smooth(x, sigma=1.0, taps=None):
    return _GaussSmooth.apply(x, float(sigma), taps)


class _UpsampleField(torch.autograd.Function):
    """Low-resolution velocity (N,d,g...) -> id + scale * F.interpolate(v, size, linear, align_corners=False)
    (adv_morph.py:464 + 111): the banded tensor-product kernel and its adjoint as one differentiable operator (the fused
    chain calls the same kernels directly)."""

    @staticmethod
    def forward(ctx, coef, tables, scale):
        coef = _dev(coef, "velocity")
        ctx.tables, ctx.scale = tables, float(scale)
        return raw_tp_interp(coef, tables, coef.shape[1], add_identity=True, scale=float(scale))

    @staticmethod
    def backward(ctx, g):
        return raw_tp_adjoint(_dev(g, "grad").contiguous(), ctx.tables, scale=ctx.scale), None, None


@_on_tensor_device
def upsample_field(coef, tables, scale):
    return _UpsampleField.apply(coef, tables, float(scale))


@_on_tensor_device
def demons_field_pair(vel, scale, tables, nsteps_rule, reduce_sumsq=None, opts=None):
    """(field(+scale * vel), field(-scale * vel)): the deformation and its approximate inverse, which one solver step
    always needs together (adv_morph.py:285-331), integrated as ONE batch [v; -v] -- half the launches, each twice the
    size (the 2D kernels of a 32-image batch are too small to fill 256 CUs).  Per sample the arithmetic is that of two
    separate calls: -(s*v) == (-s)*v exactly, so the results are bit-identical to demons_field(vel, +-scale)."""
    global _LAST_FIELD_BOUND
    _LAST_FIELD_BOUND = None
    qp, qm = _DemonsField.apply(vel, float(scale), tables, bool(nsteps_rule), reduce_sumsq, True, opts)
    if _LAST_FIELD_BOUND is not None:      # one bound for both halves (the max over the pair: still exact)
        rb, idx = _LAST_FIELD_BOUND
        qp._advchain_disp = [rb, None, qp._version, idx, _ride_key(rb, qp)]
        qm._advchain_disp = [rb, None, qm._version, idx, _ride_key(rb, qm)]
        _LAST_FIELD_BOUND = None
    return qp, qm


class _Consistency(torch.autograd.Function):
    """c_mse * S0 + c_a * SA + c_b * SB + c_kl * SKL  with  S0 = sum((P m - T m)^2), SA/SB = masked edge energies,
    SKL = sum m T' (log T' - log P)  (advchain/common/loss.py:55-79,102-220,223-249).  Differentiable w.r.t. the
    prediction logits only."""

    @staticmethod
    def forward(ctx, pred, ref, mask, coef, ref_is_prob, want_edges):
        pred, ref = _dev(pred, "pred"), _dev(ref, "reference")
        mask = None if mask is None else _dev(mask, "mask")
        N, K = pred.shape[:2]
        nd = pred.dim() - 2
        dims = _lib.dims_array(pred.shape[2:])
        mch = 1 if mask is None else mask.shape[1]
        need_grad = ctx.needs_input_grad[0]
        want_kl = coef[3] != 0.0
        R = None
        if need_grad and want_edges and K > 1:
            R = torch.empty((N, 2 * (K - 1)) + tuple(pred.shape[2:]), device=pred.device, dtype=torch.float32)
        # every output exists before the producer runs, and a failure between producer and consumer (the finisher zeroes
        # the slots again) evicts the shared accumulator instead of leaving partial sums for the next evaluation
        sums = torch.empty(4, device=pred.device, dtype=torch.float32)
        value = torch.empty((), device=pred.device, dtype=torch.float32)
        slots = _persistent_zeros("loss", (4, 64), pred.device)   # per-workgroup partials, 64 slots per sum; zeroed by the finisher
        lib = _lib.load()
        try:
            # f2 fused: one marching kernel straight from the logits, nothing saved but R (K = 2..4, rows of 4j <= 256 voxels)
            rc = lib.advchain_consistency_fused_fwd(_ptr(pred), _ptr(ref), _ptr(mask), _ptr(R), _ptr(slots), N, K, nd, dims, mch,
                                                    int(ref_is_prob), int(want_edges), int(want_kl), _stream()) if FUSED_LOSS else -2
            fused = rc != -2
            if fused:
                _lib.check(rc, "consistency_fused_fwd")
                P = D = None
            else:
                P = torch.empty_like(pred)
                D = torch.empty_like(pred)
                _lib.check(lib.advchain_consistency_fwd(_ptr(pred), _ptr(ref), _ptr(mask), _ptr(P), _ptr(D), _ptr(R),
                                                        _ptr(slots), N, K, nd, dims, mch, int(ref_is_prob),
                                                        int(want_edges), int(want_kl), _stream()), "consistency_fwd")
            _lib.check(lib.advchain_consistency_finish(_ptr(slots), _lib.float_array(coef), _ptr(sums), _ptr(value), 1,
                                                       _stream()), "consistency_finish")
        except BaseException:
            _forget_persistent(slots)
            raise
        if need_grad:
            if fused:
                ctx.save_for_backward(pred, ref, R, mask)
            else:
                ctx.save_for_backward(P, D, R, mask)
        ctx.fused = fused
        ctx.cfg = (coef, mch, int(ref_is_prob))
        ctx.mark_non_differentiable(sums)
        ctx.set_materialize_grads(False)      # (no zero tensor for the gradient of `sums`)
        return value, sums

    @staticmethod
    def backward(ctx, gloss, _gsums):
        if gloss is None:
            return None, None, None, None, None, None
        P, D, R, mask = ctx.saved_tensors              # (fused: pred, ref, R, mask)
        coef, mch, is_gt = ctx.cfg
        N, K = P.shape[:2]
        nd = P.dim() - 2
        gs = _dev(gloss.reshape(1), "grad")
        gpred = torch.empty_like(P)
        entry = _lib.load().advchain_consistency_fused_bwd if ctx.fused else _lib.load().advchain_consistency_bwd
        _lib.check(entry(_ptr(P), _ptr(D), _ptr(R), _ptr(mask), _ptr(gs), _ptr(gpred),
                         float(coef[0]), float(coef[1]), float(coef[2]), float(coef[3]),
                         is_gt, N, K, nd, _lib.dims_array(P.shape[2:]), mch, _stream()),
                   "consistency_fused_bwd" if ctx.fused else "consistency_bwd")
        return gpred, None, None, None, None, None


@_on_tensor_device
def consistency_sums(pred, ref, mask, coef, ref_is_prob=False, want_edges=True):
    """Returns (coef . sums, sums) with sums = [S_mse, S_edgeA, S_edgeB, S_kl] (device tensor, raw sums); `coef` has 3
    (no 'kl' term) or 4 entries."""
    coef = tuple(float(c) for c in coef)
    if len(coef) == 3:
        coef = coef + (0.0,)
    return _Consistency.apply(pred, ref, mask, coef, bool(ref_is_prob), bool(want_edges))
