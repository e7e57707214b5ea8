"""Host-side mirror of pyGPA/phase_unwrap.py (weighted least-squares unwrap)."""
import numpy as np

from . import _lib

DEFAULT_DTYPE = np.float64


def phase_unwrap(psi, weight=None, kmax=100, dtype=None):
    """Unwrap a wrapped phase image (phase_unwrap.py:141-208)."""
    psi = np.asarray(psi)
    plan = _lib.get_plan(psi.shape, 1, DEFAULT_DTYPE if dtype is None else dtype)
    return plan.unwrap(psi, weight, kmax=kmax)[0]


def phase_unwrap_prediff(dx, dy, weight=None, kmax=100, dtype=None):
    """Unwrap from pre-differenced gradients (phase_unwrap.py:282-350)."""
    dx = np.asarray(dx)
    dy = np.asarray(dy)
    plan = _lib.get_plan((dx.shape[0], dy.shape[1]), 1, DEFAULT_DTYPE if dtype is None else dtype)
    return plan.unwrap_prediff(dx, dy, weight, kmax=kmax)[0]


# the reference's *_ref variants are the same algorithm without the precomputed
# scaling (phase_unwrap.py:26-78, :211-279); tests/test_phase_unwrap.py asserts equality
phase_unwrap_ref = phase_unwrap
phase_unwrap_ref_prediff = phase_unwrap_prediff


def _wrapToPi(x):
    return (x + np.pi) % (2 * np.pi) - np.pi
