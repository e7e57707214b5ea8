import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + '.npz')) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope='session')
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]
    return get


def kidx_mismatch_is_tie(amp_stack, kidx_a, kidx_b, rtol):
    """True where two argmax maps differ only at (near-)ties of the amplitudes.

    amp_stack (K,N,M): candidate amplitudes; a mismatch is excused when the two
    chosen candidates' amplitudes agree to rtol * max amplitude."""
    ia = np.clip(kidx_a, 0, None)
    ib = np.clip(kidx_b, 0, None)
    aa = np.take_along_axis(amp_stack, ia[None], 0)[0]
    ab = np.take_along_axis(amp_stack, ib[None], 0)[0]
    return np.abs(aa - ab) <= rtol * amp_stack.max()


@pytest.fixture
def gpa_option():
    """set diagnostic switches of the library (gpa_set_option) for one test; afterwards every switch touched is back at the
    value it had before the test (the GPA_<NAME> environment variable's, or unset)"""
    from pygpa_amd import _lib
    before = {}

    def setter(name, value):
        before.setdefault(name, _lib.get_option(name))
        _lib.set_option(name, value)

    yield setter
    for name, value in before.items():
        _lib.set_option(name, value)
