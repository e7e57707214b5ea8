"""Drop-in for pyGPA/cuGPA.py: same names and call signatures, MI355X underneath.

The reference module is a CuPy re-spelling of four NumPy functions
(cuGPA.py:11-202); its acceptance tests (tests/test_cuGPA.py) require equality
with the NumPy functions on 'lockin' and the 0.9 px displacement bar."""
import numpy as np

from . import geometric_phase_analysis as _g


def cuGPA(image, kvec, sigma=22):
    """GPU version of optGPA (cuGPA.py:11-38)."""
    return _g.optGPA(image, kvec, sigma)


def wfr2_grad_opt(image, sigma, kx, ky, kw, kstep, grad=None):
    """cuGPA.py:41-87.  grad: None (np.gradient), 'diff' (forward differences along axis 0, axis 1 with NaN
    appended, cuGPA.py:58-62) or a callable phase -> (d/d axis 0, d/d axis 1)."""
    g = _g._grad_sweep(image, sigma, _g._sweep_list(kx, ky, kw, kstep), (kx, ky), grad, False, None)
    return {'w': g['w'], 'lockin': g['lockin'], 'grad': g['grad']}


def wfr2_grad_single(image, sigma, kx, ky, kw, kstep, grad=None):
    """Single-precision variant (cuGPA.py:90-133): no 'w' in the result."""
    g = _g._grad_sweep(image, sigma, _g._sweep_list(kx, ky, kw, kstep), (kx, ky), grad, False, np.float32)
    return {'lockin': g['lockin'], 'grad': g['grad']}


def wfr2_only_lockin(image, sigma, kvec, kw, kstep):
    """cuGPA.py:136-158."""
    return _g.wfr2_only_lockin(image, sigma, kvec[0], kvec[1], kw, kstep)


def wfr2_only_grad(image, sigma, kvec, kw, kstep, grad=None):
    """cuGPA.py:161-202."""
    return wfr2_grad_opt(image, sigma, kvec[0], kvec[1], kw, kstep, grad=grad)['grad']
