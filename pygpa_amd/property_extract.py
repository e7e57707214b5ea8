"""Host-side mirror of the phase-gradient -> Jacobian -> lattice-properties chain of
pyGPA/property_extract.py: the immediate consumer of the
``'grad'`` output of ``wfr2_grad_opt``.

Same names, argument order and return layout as the reference.  The per-pixel work -- the
weighted least squares of the phase gradients and the 2x2 singular value decomposition --
runs in libgpa_hip.so (gpa_phasegradient2J / gpa_props_from_jac, include/gpa_hip.h); the
k-vector bookkeeping on P x 2 arrays stays on the host.  No CPU fallback.

The Kerelsky fits, the double-strain decomposition and the dask per-pixel least squares of
the reference module are out of scope.
"""
import numpy as np

from . import _lib
from .geometric_phase_analysis import DEFAULT_DTYPE, calc_diff_from_isotropic
from .mathtools import periodic_average, periodic_difference


def _dt(dtype):
    return DEFAULT_DTYPE if dtype is None else dtype


def phasegradient2J(kvecs, grads, weights, nmperpixel, iso_ref=True, sort=0, dtype=None):
    """J (N, M, 2, 2) directly from phase gradients (property_extract.py:69-101).

    With iso_ref the result is relative to ``kvecs + calc_diff_from_isotropic(kvecs)``.
    As in the reference, ``sort != 0`` reorders kvecs and grads but not weights."""
    kvecs = np.asarray(kvecs, dtype=np.float64).reshape(-1, 2)
    grads = np.asarray(grads)
    angles = np.arctan2(kvecs[:, 1], kvecs[:, 0])
    if sort == 0:
        lkvecs = kvecs
        order = np.arange(3)
    else:
        order = np.argsort(sort * periodic_difference(angles, periodic_average(angles)))
        lkvecs = kvecs[order]
    plan = _lib.get_plan(grads.shape[1:3], max(len(kvecs), 2), _dt(dtype))
    if iso_ref:
        dks = calc_diff_from_isotropic(lkvecs)
        return plan.phasegradient2J(lkvecs, grads[order], weights, nmperpixel, dks=dks)
    return plan.phasegradient2J(kvecs, grads, weights, nmperpixel)


def phasegradient2Jac(kvecs, grads, weights, nmperpixel, dtype=None):
    """Jac = 1 + J from phase gradients (property_extract.py:55-66).  The identity is added
    inside the device kernels where Jac is consumed; here it is added to the returned array."""
    J = phasegradient2J(kvecs, grads, weights, nmperpixel, dtype=dtype)
    return np.eye(2, dtype=J.dtype) + J


def props_from_Jac(Jac, refangle=0., refscale=1., diff=False, dtype=None, _add_identity=False):
    """(angle, aniangle, alpha, kappa) of a lattice from the Jacobian of its transformation
    (property_extract.py:137-178): array of shape (4,) + Jac.shape[:-2]."""
    return _lib.props_from_jac(Jac, add_identity=_add_identity, refangle=refangle, refscale=refscale, diff=diff,
                               dtype=_dt(dtype))


def props_from_J(J, refangle=0., refscale=1, dtype=None):
    """props_from_Jac(J + 1) (property_extract.py:220-221)."""
    return props_from_Jac(J, refangle=refangle, refscale=refscale, dtype=dtype, _add_identity=True)


def get_initial_props(ks, standardize=False):
    """(r_k, theta_0 [deg], symmetry) of a set of k-vectors (property_extract.py:491-503);
    ``standardize`` (mathtools.standardize_ks) is not provided."""
    if standardize:
        raise NotImplementedError('standardize_ks is outside the accelerated path')
    kvecs = np.asarray(ks, dtype=np.float64).reshape(-1, 2)
    symmetry = 2 * len(kvecs)
    r_k = np.linalg.norm(kvecs, axis=1).mean()
    theta_0 = np.rad2deg(periodic_average(np.arctan2(kvecs[:, 1], kvecs[:, 0]), 2 * np.pi / symmetry))
    hexa = np.arange(-180, 180, 60)
    diffind = np.argmin(np.abs(theta_0 + hexa - np.rad2deg(np.arctan2(kvecs[0, 1], kvecs[0, 0]))))
    return r_k, theta_0 + hexa[diffind], symmetry


def calc_props_from_phasegradient(kvecs, grads, weights, nmperpixel, dtype=None):
    """Lattice properties directly from the sweep's phase gradients
    (property_extract.py:234-255): props_from_Jac(1 + J) with the base angle of kvecs added."""
    J = phasegradient2J(kvecs, grads, weights, nmperpixel, dtype=dtype)
    _, theta_0, _ = get_initial_props(kvecs)
    return props_from_J(J, refangle=theta_0, dtype=dtype)
