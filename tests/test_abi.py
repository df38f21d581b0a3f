"""CPU-side checks of the drop-in boundary: the product library loads, exports every symbol that
include/libiop_amd.h declares, and refuses to compute without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

import libiop_amd
from libiop_amd import build as iopx_build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def product():
    iopx_build.build()                      # hipcc cross-compiles gfx950 without a GPU
    return libiop_amd.Library()


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "libiop_amd.h")).read()
    return sorted(set(re.findall(r"\b(iopx_[a-z0-9_]+)\s*\(", hdr)))


def test_header_and_binding_agree():
    assert _declared_symbols() == sorted(libiop_amd.EXPORTED_SYMBOLS)


def test_library_exports_every_declared_symbol(product):
    for name in _declared_symbols():
        assert hasattr(product.c, name), name
    assert product.version() >= 100


def test_code_object_targets_gfx950_only(product):
    blob = open(product.path, "rb").read()
    assert b"gfx950" in blob
    for other in (b"gfx942", b"gfx90a", b"sm_90"):
        assert other not in blob


def test_no_cpu_fallback(product):
    if product.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(libiop_amd.NoDeviceError):
        product.additive_FFT(np.zeros((4, 3), dtype=np.uint64), libiop_amd.standard_basis(2), np.zeros(3, dtype=np.uint64))
    with pytest.raises(libiop_amd.NoDeviceError):
        product.merkle_tree([np.zeros((4, 3), dtype=np.uint64)], 2)
    with pytest.raises(libiop_amd.NoDeviceError):
        product.evaluate_next_f_i_over_entire_domain(np.zeros((4, 3), dtype=np.uint64), libiop_amd.standard_basis(2),
                                                     np.zeros(3, dtype=np.uint64), 2, np.zeros(3, dtype=np.uint64))


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "libiop_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "oracle/" not in text and "liboracle" not in text, os.path.join(dirpath, f)
