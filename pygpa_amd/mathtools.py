"""Host helpers mirrored from the reference's mathtools (scalar / bookkeeping only)."""
import numpy as np


def wrapToPi(x):
    """Wrap to [-pi, pi) with floored modulo (reference pyGPA/mathtools.py:72-75)."""
    return (x + np.pi) % (2 * np.pi) - np.pi
