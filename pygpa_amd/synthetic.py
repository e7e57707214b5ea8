"""Seeded synthetic moire / lattice images for tests and benchmarks.

The reference's tests draw their images from the third-party ``latticegen``
package with unseeded noise (reference tests/test_geometric_phase_analysis.py:25-41);
this is an independent, seeded generator of the same kind of image:

    img(r) = sum_i cos(2 pi k_i . (r + u(r))),   k_i = r_k (cos, sin)(xi0 + 60deg * i)

sampled on centred integer coordinates r.  Sign convention follows the
reference: the lattice is sampled at r + u (geometric_phase_analysis.py:250-253)
and ``-extract_displacement_field(img, ks)`` recovers u
(tests/test_geometric_phase_analysis.py:63-66).
"""
import numpy as np


def hex_kvecs(r_k=0.1, xi0=7.0, n=3):
    """n k-vectors of length r_k (cycles/pixel), 60 degrees apart, first at xi0 degrees."""
    ang = np.deg2rad(xi0 + 60.0 * np.arange(n))
    return r_k * np.stack([np.cos(ang), np.sin(ang)], axis=1)


def gaussian_bump_displacement(shape):
    """Displacement field of the reference's test fixture, scaled to ``shape``.

    u_x = 0.5 x exp(-0.5 ((x/(N/8))^2 + 1.2 (y/(M/6))^2)), u_y = 0 on centred
    coordinates (shape of tests/test_geometric_phase_analysis.py:12-17).
    """
    n, m = shape
    x = (np.arange(n) - n // 2)[:, None].astype(np.float64)
    y = (np.arange(m) - m // 2)[None, :].astype(np.float64)
    ux = 0.5 * x * np.exp(-0.5 * ((x / (n / 8.0)) ** 2 + 1.2 * (y / (m / 6.0)) ** 2))
    return np.stack([ux, np.zeros_like(ux)])


def hex_moire(shape, kvecs=None, u=None, noise=0.0, seed=0, dtype=np.float64):
    """Sum-of-cosines lattice image deformed by ``u`` (2,N,M), plus optional noise.

    ``noise`` is the standard deviation of white Gaussian noise drawn from
    ``np.random.default_rng(seed)`` and smoothed by a 3-tap [1,2,1]/4 filter
    along both axes (a cheap stand-in for the reference fixture's sigma=0.5
    Gaussian filter).
    """
    n, m = shape
    if kvecs is None:
        kvecs = hex_kvecs()
    x = (np.arange(n) - n // 2)[:, None].astype(np.float64)
    y = (np.arange(m) - m // 2)[None, :].astype(np.float64)
    if u is None:
        rx, ry = x, y
    else:
        rx, ry = x + u[0], y + u[1]
    img = np.zeros((n, m))
    for kx, ky in np.asarray(kvecs):
        img += np.cos(2 * np.pi * (kx * rx + ky * ry))
    if noise > 0:
        rng = np.random.default_rng(seed)
        nz = rng.normal(scale=noise, size=(n, m))
        nz = 0.25 * (np.roll(nz, 1, 0) + 2 * nz + np.roll(nz, -1, 0))
        nz = 0.25 * (np.roll(nz, 1, 1) + 2 * nz + np.roll(nz, -1, 1))
        img += nz
    return img.astype(dtype)


def explicit_klists(kvecs, kw, nx, ny):
    """Explicit wx-outer k-lists (wfr3-style) for the BASELINE configs whose K is
    not what the reference's square np.arange grid can produce.

    For every peak: wx in kx - kw + i*kw/2 (i < nx), wy in ky - kw + j*kw/2 (j < ny)
    when ny == nx, or ky + (j - ny//2)*kw/2 otherwise.
    """
    lists = []
    for kx, ky in np.asarray(kvecs):
        wxs = kx - kw + np.arange(nx) * kw / 2
        if ny == nx:
            wys = ky - kw + np.arange(ny) * kw / 2
        else:
            wys = ky + (np.arange(ny) - ny // 2) * kw / 2
        lists.append(np.array([(wx, wy) for wx in wxs for wy in wys]))
    return lists
