"""CPU: the C-ABI library loads and exports every symbol include/iap.h declares (no compute)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    h = open(os.path.join(ROOT, "include", "iap.h")).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    return sorted(set(re.findall(r"\b(iap_[a-z0-9_]+)\s*\(", h)))


def test_header_and_binding_agree():
    from imageanalysis_pipeline_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared()


def test_library_exports_every_declared_symbol():
    from imageanalysis_pipeline_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.fail("libiap_hip.so is not built: run __graft_entry__.build()")
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declared():
        assert hasattr(lib, name), name
    lib.iap_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.iap_version()


def test_ops_refuse_cpu_tensors():
    """the product path has no CPU fallback: host tensors must fail loudly"""
    import torch
    from imageanalysis_pipeline_amd import _lib, ops
    a = torch.zeros((1, 1, 1, 64, 64), dtype=torch.uint16)
    with pytest.raises(_lib.IapError):
        ops.im_reg(a)
    with pytest.raises(_lib.IapError):
        ops.bg_smo(a, 0.05)
    with pytest.raises(_lib.IapError):
        ops.measure_accumulate(a, a[:, 0])


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "imageanalysis_pipeline_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert "oracle" not in src.replace("no oracle", ""), f
