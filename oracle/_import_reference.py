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
    # the reference's top-level ``advchain`` directory has no __init__.py (namespace package), so a regular
    # package of the same name on sys.path (our alias) would shadow it: bind the name explicitly.
    pkg = types.ModuleType("advchain")
    pkg.__path__ = [os.path.join(REFERENCE_ROOT, "advchain")]
    sys.modules["advchain"] = pkg
    import importlib
    aug = importlib.import_module("advchain.augmentor")
    importlib.import_module("advchain.common.loss")
    importlib.import_module("advchain.common.utils")
    assert aug.__file__.startswith(REFERENCE_ROOT), aug.__file__
    return aug
