"""CPU-side checks of the C-ABI boundary: the library loads, exports every symbol
include/gpa_hip.h declares, and refuses loudly to run without a GPU."""
import os
import re

import numpy as np
import pytest

from pygpa_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ensure_built():
    if not os.path.exists(_lib.LIB_PATH):
        from pygpa_amd import build
        build.build(verbose=False)


def test_header_symbols_exported():
    _ensure_built()
    lib = _lib.load()
    header = open(os.path.join(ROOT, 'include', 'gpa_hip.h')).read()
    header = re.sub(r'/\*.*?\*/', '', header, flags=re.S)
    names = sorted(set(re.findall(r'\b(gpa_[A-Za-z0-9_]+)\s*\(', header)))
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), 'symbol %s declared in gpa_hip.h is not exported' % n
        assert n in _lib.SIGNATURES, 'symbol %s has no ctypes prototype' % n
    assert sorted(_lib.SIGNATURES) == names


def test_version_and_error_string():
    _ensure_built()
    lib = _lib.load()
    assert lib.gpa_version() >= 100
    assert lib.gpa_plan_create(0, 2, 2, 1, 0) is None     # too small: rejected before touching a device
    assert 'gpa_plan_create' in _lib.last_error()


def test_no_cpu_fallback():
    """Without a GPU the product path must raise, never compute on the host."""
    _ensure_built()
    lib = _lib.load()
    if lib.gpa_device_count() > 0:
        pytest.skip('a GPU is visible')
    import pygpa_amd.geometric_phase_analysis as g
    img = np.zeros((64, 64))
    with pytest.raises(_lib.GPAError):
        g.optGPA(img, (0.1, 0.0), 5)
    with pytest.raises(_lib.GPAError):
        g.extract_displacement_field(img, np.array([[0.1, 0], [0, 0.1], [0.07, 0.07]]))


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: no module of the product package may reference it."""
    pkg = os.path.join(ROOT, 'pygpa_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), f
                assert 'gpa_oracle' not in src, f
