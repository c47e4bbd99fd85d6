"""Import recipe for the upstream reference (THIS CONTAINER ONLY; never runs on the GPU box).

TEST INFRASTRUCTURE.  Used by oracle/make_golden.py (fixture generator) and by the
optional "oracle vs live reference" CPU tests, which skip when /root/reference is absent.
The recipe is the one recorded in SURVEY.md Appendix C: two unused third-party imports
are stubbed (cv2: adv_bias.py:1, SimpleITK: common/utils.py:7) and np.Inf is aliased
(adv_bias.py:237-238 under NumPy 2).
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("ADVCHAIN_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "advchain", "augmentor"))


def import_reference():
    """Returns the reference's ``advchain.augmentor`` module (imported from REFERENCE_ROOT)."""
    import numpy as np
    if not reference_available():
        raise RuntimeError("reference not mounted at %s" % REFERENCE_ROOT)
    for name in ("cv2", "SimpleITK"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.magnitude = None
            sys.modules[name] = m
    if not hasattr(np, "Inf"):
        np.Inf = np.inf
    # our repo ships a drop-in alias package also called ``advchain``; make sure the
    # reference's one wins inside this process.
    for k in [k for k in sys.modules if k == "advchain" or k.startswith("advchain.")]:
        del sys.modules[k]
    sys.path.insert(0, REFERENCE_ROOT)
    try:
        import advchain.augmentor as aug  # noqa
        import advchain.common.loss as loss  # noqa
        import advchain.common.utils as utils  # noqa
    finally:
        sys.path.remove(REFERENCE_ROOT)
    return aug
