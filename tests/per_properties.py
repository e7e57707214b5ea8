"""Moisan (2011) periodic-plus-smooth decomposition u = p + s, checked WITHOUT the formula the oracle and the device
kernels implement (the k-space division p_hat = u_hat - v_hat / (2 cos + 2 cos - 4)): helpers shared by the CPU test of
the oracle's `per` and the GPU test of gpa_per_dft.  Spatial domain only:

  (1) Moisan's theorem 1: s is the zero-mean solution of  Lap_per s = v,  Lap_per the 4-neighbour Laplacian with
      PERIODIC neighbours, v the image of the jumps across the border of u -- so s is discretely harmonic away from the
      border (Lap_per s = 0 there) and its Laplacian ON the border is the jump of u across it;
  (2) the variational statement the decomposition is defined by: among all splittings u = p + s with mean(s) = 0, this
      one minimises  E = (squared jumps of p across the periodic border) + (squared interior differences of s).
      E is a convex quadratic, so "every zero-mean perturbation raises E" is checked on random directions and on the
      directions that concentrate at the border.
"""
import numpy as np


def border_jump_image(u):
    v = np.zeros_like(u, dtype=np.float64)
    v[0, :] += u[-1, :] - u[0, :]
    v[-1, :] += u[0, :] - u[-1, :]
    v[:, 0] += u[:, -1] - u[:, 0]
    v[:, -1] += u[:, 0] - u[:, -1]
    return v


def periodic_laplacian(s):
    return np.roll(s, 1, 0) + np.roll(s, -1, 0) + np.roll(s, 1, 1) + np.roll(s, -1, 1) - 4 * s


def energy(p, s):
    cross = ((p[0, :] - p[-1, :]) ** 2).sum() + ((p[:, 0] - p[:, -1]) ** 2).sum()
    inner = ((s[1:, :] - s[:-1, :]) ** 2).sum() + ((s[:, 1:] - s[:, :-1]) ** 2).sum()
    return cross + inner


def check_decomposition(u, phat, tol, rng):
    """u: image, phat: claimed DFT of its periodic component.  Returns the residuals of (1); asserts (1) and (2)."""
    u = np.asarray(u, dtype=np.float64)
    p_c = np.fft.ifft2(np.asarray(phat, dtype=np.complex128))
    scale = max(np.abs(u).max(), 1e-300)
    assert np.abs(p_c.imag).max() < tol * scale, 'the periodic component of a real image must be real'
    p = p_c.real
    s = u - p
    assert abs(s.mean()) < tol * scale
    res = periodic_laplacian(s) - border_jump_image(u)
    assert np.abs(res).max() < 8 * tol * scale, 'Lap_per s != border jumps of u (max %.3e)' % np.abs(res).max()
    inner = res[1:-1, 1:-1]
    assert np.abs(inner).max() < 8 * tol * scale       # harmonic in the interior
    e0 = energy(p, s)
    n0, n1 = u.shape
    dirs = [rng.standard_normal(u.shape) for _ in range(4)]
    edge = np.zeros(u.shape)
    edge[0, :] = 1.0
    edge[:, -1] = -1.0
    dirs.append(edge)
    for d in dirs:
        d = d - d.mean()
        for eps in (1e-2, -1e-2):
            assert energy(p + eps * scale * d, s - eps * scale * d) > e0 - 64 * tol * scale * scale * u.size
    return res
