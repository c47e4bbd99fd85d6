"""The C-ABI library builds for gfx950 (hipcc cross-compiles without a GPU), loads, and exports every symbol
that include/advchain_hip.h declares; the ctypes prototype table covers exactly that set.  No compute calls."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "advchain_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(advchain_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    syms = declared_symbols()
    for must in ("advchain_grid_sample_fwd", "advchain_grid_sample_bwd", "advchain_compose_self_fwd",
                 "advchain_compose_self_bwd", "advchain_affine_warp_fwd", "advchain_affine_warp_bwd",
                 "advchain_bias_field_fwd", "advchain_bias_field_bwd", "advchain_gauss_axis",
                 "advchain_consistency_fwd", "advchain_consistency_bwd", "advchain_norm_axpy"):
        assert must in syms


def test_library_builds_loads_and_exports_every_declared_symbol():
    from advchain_amd.build import build_library
    path = build_library()
    assert os.path.exists(path)
    cdll = ctypes.CDLL(path)
    for name in declared_symbols():
        assert hasattr(cdll, name), "missing export: " + name


def test_ctypes_prototypes_match_the_header():
    from advchain_amd import _lib
    assert sorted(_lib.PROTOTYPES) == declared_symbols()
    lib = _lib.load()
    assert lib.advchain_version() >= 100
    # argument-validation paths run on the host (no kernel launch, no GPU needed)
    rc = lib.advchain_axpy(None, None, None, 1.0, 8, None)
    assert rc < 0 and b"axpy" in lib.advchain_last_error()


def test_product_has_no_cpu_fallback():
    import pytest
    import torch
    from advchain_amd import _lib, ops
    from advchain_amd.augmentor import AdvNoise
    with pytest.raises(_lib.AdvchainHipError):
        ops.grid_sample(torch.rand(1, 1, 4, 4), torch.rand(1, 2, 4, 4))
    t = AdvNoise(spatial_dims=2, config_dict=dict(epsilon=0.1, xi=1e-6, data_size=[1, 1, 4, 4]),
                 device=torch.device("cpu"))
    with pytest.raises(_lib.AdvchainHipError):
        t.forward(torch.rand(1, 1, 4, 4))


def test_product_never_imports_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/."""
    for base, _, files in os.walk(os.path.join(ROOT, "advchain_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(base, f)).read()
                assert "oracle" not in src.replace("the CPU oracle", ""), os.path.join(base, f)
