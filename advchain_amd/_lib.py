"""ctypes binding of ``libadvchain_hip.so`` (the C ABI declared in ``include/advchain_hip.h``).

There is deliberately NO fallback: if the shared library is missing or a call fails, an exception is
raised.  The product path never routes through PyTorch reference ops or the CPU oracle.
"""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libadvchain_hip.so")

_P, _I, _L, _F = c_void_p, c_int, c_int64, c_float

# name -> (restype, argtypes).  Must list every symbol of include/advchain_hip.h.
PROTOTYPES = {
    "advchain_version": (_I, []),
    "advchain_last_error": (c_char_p, []),
    "advchain_set_deterministic": (None, [_I]),
    "advchain_get_deterministic": (_I, []),
    "advchain_grid_sample_fwd": (_I, [_P, _P, _P, _L, _L, _I, _P, _P, _I, _I, _I, _P]),
    "advchain_grid_sample_fwd_ride": (_I, [_P, _P, _P, _P, _P, _L, _L, _I, _P, _P, _I, _I, _I, _I, _P]),
    "advchain_scatter_workspace": (_L, [_L, _I, _P]),
    "advchain_grid_sample_bwd": (_I, [_P, _P, _P, _P, _P, _P, _L, _L, _I, _P, _P, _I, _I, _I, _I, _P]),
    "advchain_compose_self_fwd": (_I, [_P, _P, _P, _L, _I, _P, _I, _P, _P]),
    "advchain_compose_self_bwd": (_I, [_P, _P, _P, _P, _I, _I, _L, _I, _P, _P]),
    "advchain_affine_warp_fwd": (_I, [_P, _P, _P, _L, _L, _I, _P, _I, _I, _P]),
    "advchain_affine_warp_fwd_ride": (_I, [_P, _P, _P, _P, _P, _L, _L, _I, _P, _I, _I, _I, _P]),
    "advchain_affine_warp_bwd_workspace": (_L, [_L, _I, _P]),
    "advchain_affine_warp_bwd": (_I, [_P, _P, _P, _P, _P, _P, _L, _L, _I, _P, _I, _I, _P]),
    "advchain_affine_theta_fwd": (_I, [_P, _P, _F, _P, _P, _L, _I, _P]),
    "advchain_affine_theta_bwd": (_I, [_P, _P, _F, _P, _P, _P, _L, _I, _P]),
    "advchain_tp_interp_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _L, _L, _I, _I, _F, _P, _P, _P]),
    "advchain_tp_interp_fwd_smoothed_pair": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _L, _L, _I, _F, _P, _P, _F, _P]),
    "advchain_band_reduce_axis": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _L, _L, _F, _P]),
    "advchain_band_reduce_rows_dense": (_I, [_P, _P, _P, _P, _P, _L, _L, _L, _L, _F, _P]),
    "advchain_bias_field_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _L, _F, _I, _F, _P]),
    "advchain_bias_field_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _L, _F, _I, _F, _P]),
    "advchain_gauss_axis": (_I, [_P, _P, _P, _L, _L, _I, _P, _I, _P, _I, _I, _F, _P]),
    "advchain_gauss_axis_generic": (_I, [_P, _P, _L, _I, _P, _I, _P, _I, _F, _P]),
    "advchain_max_displacement": (_I, [_P, _P, _L, _I, _P, _P]),
    "advchain_grid_sample_bicubic2d_fwd": (_I, [_P, _P, _P, _L, _L, _P, _P, _I, _P]),
    "advchain_grid_sample_bicubic2d_bwd": (_I, [_P, _P, _P, _P, _P, _L, _L, _P, _P, _I, _P]),
    "advchain_affine_grid2d_fwd": (_I, [_P, _P, _L, _P, _P]),
    "advchain_affine_grid2d_bwd_workspace": (_L, [_L, _P]),
    "advchain_affine_grid2d_bwd": (_I, [_P, _P, _P, _L, _P, _P]),
    "advchain_slot_rows_max": (_I, [_P, _P, _L, _L, _I, _P]),
    "advchain_demons_compose_pair_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _P, _P, _F, _F, _I, _P]),
    "advchain_demons_compose_pair_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _P, _P, _P, _L,
                                              _I, _I, _P, _F, _F, _P]),
    "advchain_bounds_check": (_I, [_P, _P, _P, _L, _P, _P]),
    "advchain_expo_chain_fused_levels": (_I, [_L, _I, _P, _I, _P]),
    "advchain_expo_chain_fwd": (_I, [_P, _P, _P, _L, _I, _P, _I, _P, _P, _P, _P]),
    "advchain_expo_chain_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _L, _I, _P, _I, _P]),
    "advchain_gauss_xy": (_I, [_P, _P, _P, _L, _L, _I, _P, _P, _I, _I, _F, _P, _P, _L]),
    "advchain_gauss_small": (_I, [_P, _P, _L, _I, _P, _P, _I, _F, _P]),
    "advchain_gauss_small_pair": (_I, [_P, _P, _L, _I, _P, _P, _F, _I, _P]),
    "advchain_axpy": (_I, [_P, _P, _P, _F, _L, _P]),
    "advchain_sign_axpy": (_I, [_P, _P, _P, _F, _L, _P, _P, _P]),
    "advchain_nonzero_mask": (_I, [_P, _P, _L, _P]),
    "advchain_consistency_finish": (_I, [_P, _P, _P, _P, _I, _P]),
    "advchain_norm_workspace": (_L, [_L, _L]),
    "advchain_norm_axpy": (_I, [_P, _P, _P, _P, _F, _L, _L, _P]),
    "advchain_norm_axpy_gated": (_I, [_P, _P, _P, _P, _F, _L, _L, _P, _P, _P]),
    "advchain_update_multi": (_I, [_P, _I, _P, _P]),
    "advchain_consistency_fwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _L, _L, _I, _P, _I, _I, _I, _I, _P]),
    "advchain_consistency_bwd": (_I, [_P, _P, _P, _P, _P, _P, _F, _F, _F, _F, _I, _L, _L, _I, _P, _I, _P]),
    "advchain_consistency_fused_fwd": (_I, [_P, _P, _P, _P, _P, _L, _L, _I, _P, _I, _I, _I, _I, _P]),
    "advchain_consistency_fused_bwd": (_I, [_P, _P, _P, _P, _P, _P, _F, _F, _F, _F, _I, _L, _L, _I, _P, _I, _P]),
    "advchain_consistency_fused_fwd_bf16": (_I, [_P, _P, _P, _P, _P, _L, _L, _I, _P, _P]),
    "advchain_consistency_fused_bwd_bf16": (_I, [_P, _P, _P, _P, _P, _P, _F, _F, _F, _L, _L, _I, _P, _P]),
}

class UpdateDesc(ctypes.Structure):
    """advchain_update_desc of include/advchain_hip.h."""
    _fields_ = [("base", c_void_p), ("x", c_void_p), ("out", c_void_p), ("old", c_void_p), ("N", c_int64), ("M", c_int64),
                ("kind", ctypes.c_int32), ("step", c_float)]


_lib = None


class AdvchainHipError(RuntimeError):
    pass


class _Lib(object):
    """Thin proxy over the ctypes library.  ``timed`` (bench / profiling only) wraps selected entry points with
    a pair of events on the current stream -- the stream the kernels are launched on -- and records
    (name, args, start_event, end_event); everything else is a direct ctypes call."""

    def __init__(self, cdll):
        self._cdll = cdll
        self._raw = {}
        self.records = []

    def _bind(self, name, res, args):
        fn = getattr(self._cdll, name)  # AttributeError if the symbol is missing: fail loudly
        fn.restype = res
        fn.argtypes = args
        self._raw[name] = fn
        setattr(self, name, fn)

    def timed(self, names=None, every=1):
        """Event pairs (on the current stream) around the chosen entry points; `every` = k samples one call in k."""
        import contextlib
        import torch

        @contextlib.contextmanager
        def ctx():
            chosen = [n for n in self._raw if (names is None or n in names) and self._raw[n].restype is c_int
                      and len(self._raw[n].argtypes) > 0]
            for n in chosen:
                raw = self._raw[n]

                count = [0]

                def wrapped(*a, _raw=raw, _n=n, _count=count):
                    _count[0] += 1
                    if _count[0] % every:
                        return _raw(*a)
                    e0 = torch.cuda.Event(enable_timing=True)
                    e1 = torch.cuda.Event(enable_timing=True)
                    e0.record()
                    rc = _raw(*a)
                    e1.record()
                    self.records.append((_n, a, e0, e1))
                    return rc
                setattr(self, n, wrapped)
            try:
                yield self
            finally:
                for n in chosen:
                    setattr(self, n, self._raw[n])
        return ctx()


def load():
    """Loads the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AdvchainHipError(
            "libadvchain_hip.so not found at %s -- build it with `python -m advchain_amd.build` "
            "(there is no CPU / PyTorch fallback for the advchain_amd kernels)" % LIB_PATH)
    # make sure the HIP runtime PyTorch uses is the one already mapped (same SONAME libamdhip64.so.7)
    import torch  # noqa: F401
    tl = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(tl):
        ctypes.CDLL(tl, mode=ctypes.RTLD_GLOBAL)
    lib = _Lib(ctypes.CDLL(LIB_PATH))
    for name, (res, args) in PROTOTYPES.items():
        lib._bind(name, res, args)
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().advchain_last_error()
        raise AdvchainHipError("%s failed (rc=%d): %s" % (what, rc, msg.decode() if msg else ""))


_DIMS = {}


def dims_array(dims):
    key = tuple(dims)
    arr = _DIMS.get(key)
    if arr is None:
        arr = _DIMS[key] = (c_int64 * len(key))(*[int(d) for d in key])   # read-only for the library: safe to share
    return arr


def float_array(vals):
    return (c_float * len(vals))(*[float(v) for v in vals])
