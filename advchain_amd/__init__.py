"""advchain_amd -- MI355X-native implementation of AdvChain's adversarial-augmentation inner loop.

Python host code (same ``advchain.augmentor`` API as the reference) over hand-written HIP kernels for
gfx950, reached through a C ABI (``include/advchain_hip.h``, ``advchain_amd/csrc``).  There is no CPU or
PyTorch-op fallback: the kernels need a ROCm device and the built ``libadvchain_hip.so``.
"""
__version__ = "0.1.0"

from . import _lib  # noqa: F401


def library_path():
    return _lib.LIB_PATH
