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
    # ... and nothing else: the library is built with -fvisibility=hidden, its dynamic symbol table is the header
    import shutil
    import subprocess
    nm = shutil.which('nm') or shutil.which('llvm-nm') or '/opt/rocm/lib/llvm/bin/llvm-nm'
    if os.path.exists(nm):
        out = subprocess.run([nm, '-D', '--defined-only', _lib.LIB_PATH], capture_output=True, text=True).stdout
        exported = sorted({ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-2] in ('T', 'W', 'D', 'B', 'V')})
        own = [e for e in exported if not e.startswith('__hip') and not e.startswith('_fini') and not e.startswith('_init')]
        # (hipcc gives the host-side handles of __global__ kernels default visibility whatever the flag: mangled *_kernel names)
        kernels = [e for e in own if e.startswith('_Z') and 'kernel' in e]
        # (... and libstdc++'s own templates are declared with default visibility by its headers: weak std:: instantiations)
        std = [e for e in own if re.match(r'_Z(Z?N?K?St|N9__gnu_cxx|T[ISV]|GV)', e)]
        rest = sorted(set(own) - set(kernels) - set(std))
        assert rest == names, sorted(set(rest) ^ set(names))[:20]


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


def test_device_buffer_selects_its_device_first(monkeypatch):
    """hipMalloc / hipMemcpy act on the calling thread's current device, so a DeviceBuffer of a plan on
    GPU n must select GPU n before every runtime call -- also from a fresh thread (the upload / download workers of
    the stack call), whose current device is 0.  Checked against a recording stand-in for the HIP runtime (the test
    boxes have one GPU)."""
    import ctypes as C
    import threading
    _ensure_built()
    _lib.load()
    calls = []

    class FakeHip:
        def hipSetDevice(self, d):
            calls.append(('set', threading.get_ident(), d.value))
            return 0

        def hipMalloc(self, pp, n):
            calls.append(('malloc', threading.get_ident(), n.value))
            C.cast(pp, C.POINTER(C.c_void_p))[0] = 0x1000
            return 0

        def hipMemcpy(self, dst, src, n, kind):
            calls.append(('memcpy', threading.get_ident(), kind))
            return 0

        def hipFree(self, p):
            calls.append(('free', threading.get_ident(), p.value))
            return 0

    monkeypatch.setattr(_lib.DeviceBuffer, '_hip', FakeHip())
    buf = _lib.DeviceBuffer(64, device=3)
    assert [c[0] for c in calls] == ['set', 'malloc'] and calls[0][2] == 3

    def worker():
        buf.upload(np.zeros(8))
        buf.download((8,), np.float64)
        buf.download_into(np.zeros(8))
    t = threading.Thread(target=worker)
    t.start()
    t.join()
    buf.free()
    ops = [c[0] for c in calls[2:]]
    assert ops == ['set', 'memcpy', 'set', 'memcpy', 'set', 'memcpy', 'set', 'free']
    assert all(c[2] == 3 for c in calls if c[0] == 'set')
    # the worker thread made its own selections
    assert {c[1] for c in calls[2:8]} != {calls[0][1]}


def test_options_restore_previous_values(monkeypatch):
    """`options()` (and the gpa_option fixture) put every switch back to the value it had -- an outer block's, the
    GPA_<NAME> environment variable's, or unset -- instead of clearing it"""
    _ensure_built()
    monkeypatch.setenv('GPA_TRI_Q', '2')
    _lib._OPTION_VALUES.pop('TRI_Q', None)
    _lib._OPTION_VALUES.pop('COLSOLVE', None)
    with _lib.options(TRI_Q=None, COLSOLVE='tri'):
        assert _lib.get_option('TRI_Q') is None and _lib.get_option('COLSOLVE') == 'tri'
        with _lib.options(COLSOLVE='fft'):
            assert _lib.get_option('COLSOLVE') == 'fft'
        assert _lib.get_option('COLSOLVE') == 'tri'
    assert _lib.get_option('TRI_Q') == '2' and _lib.get_option('COLSOLVE') is None
    _lib.set_option('TRI_Q', None)
    with pytest.raises(_lib.GPAError):
        _lib.set_option('NO_SUCH_SWITCH', '1')
    _lib._OPTION_VALUES.pop('NO_SUCH_SWITCH', None)
