"""Host helpers mirrored from the reference's mathtools (scalar / bookkeeping only)."""
import numpy as np


def wrapToPi(x):
    """Wrap to [-pi, pi) with floored modulo (reference pyGPA/mathtools.py:72-75)."""
    return (x + np.pi) % (2 * np.pi) - np.pi


def fit_plane(image, verbose=False):
    """Huber-loss plane fit a[0]*x + a[1]*y + a[2] through `image` (reference
    pyGPA/mathtools.py:30-47).  A 3-parameter robust fit: host-side SciPy in the reference
    and here ("Huber fit is host-side SciPy (stays on host)")."""
    import scipy.optimize as spo
    lxx, lyy = np.meshgrid(np.arange(image.shape[0]), np.arange(image.shape[1]), indexing='ij')

    def resid(x):
        return (image - (x[0] * lxx + x[1] * lyy + x[2])).ravel()
    res = spo.least_squares(resid, np.zeros(3), loss='huber')
    if verbose:
        print(res.message)
    return res.x


def periodic_average(X, period=2 * np.pi, weights=1., **kwargs):
    """Periodic (circular) mean of X (mathtools.py:6-10)."""
    Y = np.angle((weights * np.exp(2j * np.pi / period * np.asarray(X))).mean(**kwargs))
    return Y * period / (2 * np.pi)


def periodic_difference(X, Y, period=2 * np.pi):
    """Periodic difference X - Y folded into [-period/2, period/2) (mathtools.py:13-17)."""
    return np.angle(np.exp(2j * np.pi / period * (np.asarray(X) - np.asarray(Y)))) * period / (2 * np.pi)
