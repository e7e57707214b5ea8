"""Host helpers mirrored from the reference's mathtools (scalar / bookkeeping only)."""
import numpy as np


def wrapToPi(x):
    """Wrap to [-pi, pi) with floored modulo (reference pyGPA/mathtools.py:72-75)."""
    return (x + np.pi) % (2 * np.pi) - np.pi


def fit_plane(image, verbose=False, dtype=None):
    """Huber-loss plane fit a[0]*x + a[1]*y + a[2] through `image` (x, y = row, column index;
    reference pyGPA/mathtools.py:30-47).  The reference runs scipy.optimize.least_squares with
    loss='huber'; here the same convex cost is minimised by iteratively reweighted least squares
    whose per-pass sums are reduced on the device (gpa_fit_plane, include/gpa_hip.h)."""
    from . import _lib
    image = np.asarray(image)
    if image.ndim != 2:
        raise ValueError('image must be 2-D')
    if dtype is None:
        dtype = np.float32 if image.dtype == np.float32 else np.float64
    coef, iters = _lib.get_plan(image.shape, 1, dtype).fit_plane(image)
    if verbose:
        print('IRLS Huber plane fit: %d passes' % iters)
    return coef


def periodic_average(X, period=2 * np.pi, weights=1., **kwargs):
    """Periodic (circular) mean of X (mathtools.py:6-10)."""
    Y = np.angle((weights * np.exp(2j * np.pi / period * np.asarray(X))).mean(**kwargs))
    return Y * period / (2 * np.pi)


def periodic_difference(X, Y, period=2 * np.pi):
    """Periodic difference X - Y folded into [-period/2, period/2) (mathtools.py:13-17)."""
    return np.angle(np.exp(2j * np.pi / period * (np.asarray(X) - np.asarray(Y)))) * period / (2 * np.pi)


def remove_negative_duplicates(ks):
    """One representative per +k / -k pair (mathtools.py:78-94): every vector is oriented so that its
    first non-zero coordinate is positive, then a vector is kept unless it coincides with an earlier
    kept one within np.isclose's tolerance (rtol 1e-5 of the kept vector, atol = 1e-3 of the mean
    smaller |coordinate|)."""
    ks = np.asarray(ks)
    if len(ks) == 0:
        return ks
    lead = np.where(ks[:, 0] != 0, ks[:, 0], ks[:, 1])
    oriented = ks * np.sign(lead)[:, None]
    atol = 1e-3 * np.abs(oriented).min(axis=1).mean()
    kept = np.empty_like(oriented)
    n = 0
    for k in oriented:
        near = np.abs(k - kept[:n]) <= atol + 1e-5 * np.abs(kept[:n])
        if not near.all(axis=1).any():
            kept[n] = k
            n += 1
    return kept[:n].copy()
