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
    """cuGPA.py:41-87.  Only the default np.gradient stencil (grad=None) is provided."""
    if grad is not None:
        raise NotImplementedError("only grad=None (np.gradient stencil) is provided")
    g = _g.wfr2_grad_opt(image, sigma, kx, ky, kw, kstep)
    return {'w': g['w'], 'lockin': g['lockin'], 'grad': g['grad']}


def wfr2_grad_single(image, sigma, kx, ky, kw, kstep, grad=None):
    """Single-precision variant (cuGPA.py:90-133): no 'w' in the result."""
    if grad is not None:
        raise NotImplementedError("only grad=None (np.gradient stencil) is provided")
    g = _g.wfr2_grad_opt(image, sigma, kx, ky, kw, kstep, dtype=np.float32)
    return {'lockin': g['lockin'], 'grad': g['grad']}


def wfr2_only_lockin(image, sigma, kvec, kw, kstep):
    """cuGPA.py:136-158."""
    return _g.wfr2_only_lockin(image, sigma, kvec[0], kvec[1], kw, kstep)


def wfr2_only_grad(image, sigma, kvec, kw, kstep, grad=None):
    """cuGPA.py:161-202."""
    return wfr2_grad_opt(image, sigma, kvec[0], kvec[1], kw, kstep, grad=grad)['grad']
