"""CPU oracle for the GPA hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A NumPy/SciPy restatement of the windowed-Fourier lock-in displacement-field
pipeline of TAdeJong/pyGPA.  It is the checker the
HIP path is compared against and the ``cpu_baseline`` leg of ``bench.py``.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline may
import this module; the product package ``pygpa_amd`` never does.

Parity status: PINNED for a1..a8, f-1 (Lawler-Fujita), f-2 (Jacobian / lattice properties), f-4 (Huber
plane fit = the reference's SciPy call) and the driver logic + smoothing of f-3 (peak finding, with
declared stand-ins for moisan2011.per and skimage.feature.peak_local_max) -- every function below is checked against
outputs of the real reference (imported from /root/reference in the build
container by ``oracle/make_golden.py``; vectors committed under
``tests/golden/``) by ``tests/test_oracle_golden.py``.
UNPINNED for a9 (``per``): the arithmetic lives in the third-party package
``moisan2011`` (unpinned git HEAD of github.com/TAdeJong/moisan2011, named at
reference README.md:18,24 and .github/workflows/ci.yaml:32), which is absent
from /root/reference; it is restated from Moisan (2011) and self-checked with
known answers only.

All ``file:line`` citations are into the reference checkout
(``pyGPA/geometric_phase_analysis.py`` = gpa.py, ``pyGPA/phase_unwrap.py`` =
pu.py, ``pyGPA/mathtools.py`` = mt.py, ``pyGPA/cuGPA.py`` = cu.py).

Conventions: image axis 0 <-> "x" <-> kvec[0], axis 1
<-> "y" <-> kvec[1]; forward FFT unnormalised, inverse 1/(N*M).
"""
import numpy as np
import scipy.fft as sfft
import scipy.optimize as spo

TWO_PI = 2.0 * np.pi


# --------------------------------------------------------------------------
# small helpers
# --------------------------------------------------------------------------
def wrap_to_pi(x):
    """(x + pi) mod 2pi - pi with floored mod: range [-pi, pi), +pi -> -pi.

    Follows mt.py:72-75 (wrapToPi) and pu.py:135-138 (_wrapToPi).
    """
    return np.mod(x + np.pi, TWO_PI) - np.pi


def gaussian_kspace_1d(n, sigma):
    """1-D factor of the k-space Gaussian: exp(-2 pi^2 sigma^2 f^2), f = fftfreq(n).

    ``scipy.ndimage.fourier_gaussian`` (called at gpa.py:44, :75, :87 and
    cu.py:57) is the outer product of two of these, each factor flushed to
    exactly 0 where its exponent exceeds 50 (SciPy's ni_fourier.c does that per
    axis); checked against SciPy in tests/test_oracle_golden.py.
    """
    f = np.fft.fftfreq(n)
    e = 2.0 * np.pi ** 2 * sigma ** 2 * f * f
    return np.where(e > 50.0, 0.0, np.exp(-e))


def carrier_1d(n, k):
    """exp(2 pi i * x * k) for x = 0..n-1 (one separable factor of gpa.py:73)."""
    return np.exp(2j * np.pi * np.arange(n) * k)


# --------------------------------------------------------------------------
# a1 / a2: single and batched lock-in
# --------------------------------------------------------------------------
def lockin(image, kvec, sigma, workers=1):
    """One spatial lock-in: ifft2(fft2(image * carrier) * Gaussian).

    Follows GPA (gpa.py:20-45), optGPA (gpa.py:48-76), cuGPA (cu.py:11-38).
    Always returns complex128 (the reference's carrier is complex128).
    """
    image = np.asarray(image)
    n0, n1 = image.shape
    carrier = carrier_1d(n0, kvec[0])[:, None] * carrier_1d(n1, kvec[1])[None, :]
    spec = sfft.fft2(image * carrier, workers=workers)
    spec *= gaussian_kspace_1d(n0, sigma)[:, None]
    spec *= gaussian_kspace_1d(n1, sigma)[None, :]
    return sfft.ifft2(spec, workers=workers)


def lockin_batch(image, kvecs, sigma, workers=1):
    """Batched lock-in over a (B, 2) list of k-vectors -> (B, N, M).

    Follows vecGPA (gpa.py:79-89).
    """
    kvecs = np.atleast_2d(np.asarray(kvecs, dtype=np.float64))
    return np.stack([lockin(image, kv, sigma, workers=workers) for kv in kvecs])


# --------------------------------------------------------------------------
# a3 / a4: reference-vector sweep
# --------------------------------------------------------------------------
def sweep_grid(kx, ky, kw, kstep):
    """The (K, 2) list of reference vectors of the reference's double loop,
    wx outer / wy inner, built with np.arange exactly as gpa.py:679-680 does
    (its length is float-rounding dependent, so it is always built on the host).
    """
    wxs = np.arange(kx - kw, kx + kw, kstep)
    wys = np.arange(ky - kw, ky + kw, kstep)
    return np.array([(wx, wy) for wx in wxs for wy in wys], dtype=np.float64).reshape(-1, 2)


def sweep(image, sigma, klist, kref, want_grad=False, workers=1, pool=1):
    """Adaptive lock-in over an explicit k-list.

    For every k in ``klist`` (in order) compute sf = lockin(image, k); a pixel
    takes the candidate when |sf| is strictly larger than the amplitude kept so
    far (accumulator starts at 0).  The stored value is re-referenced to
    ``kref``: sf * exp(-2 pi i ((wx-kx) x + (wy-ky) y)).

    Follows optwfr2 (gpa.py:669-686), wfr2 (gpa.py:615-644), wfr3
    (gpa.py:647-666), wfr2_only_lockin (gpa.py:689-702) and, with
    ``want_grad``, wfr2_grad_opt (gpa.py:763-813; GPU twin cu.py:41-87).

    Returns dict: 'lockin' (N,M) c128, 'kidx' (N,M) int32 index into klist of
    the winner (-1 where no candidate ever won), 'w' (2,N,M) f64 (gpa.py:685),
    and with want_grad 'grad' (N,M,2) f64.

    ``pool`` > 1 computes the candidates' lock-ins `pool` at a time on a thread pool (the
    analogue of the reference's dask-vectorised wfr2_only_lockin_vec, gpa.py:705-719; NumPy's
    FFT and elementwise loops release the GIL); the selection below still runs in list
    order, so the result is the same.  Only bench.py's multi-core CPU baseline uses it.
    """
    image = np.asarray(image)
    klist = np.asarray(klist, dtype=np.float64).reshape(-1, 2)
    n0, n1 = image.shape
    best = np.zeros((n0, n1), dtype=np.complex128)
    best_amp = np.zeros((n0, n1))
    kidx = np.full((n0, n1), -1, dtype=np.int32)
    grad = np.zeros((n0, n1, 2)) if want_grad else None
    ahead = {}
    executor = None
    if pool > 1:
        from concurrent.futures import ThreadPoolExecutor
        executor = ThreadPoolExecutor(max_workers=pool)
    for i, (wx, wy) in enumerate(klist):
        if executor is not None:
            if i % pool == 0:
                ahead = {j: executor.submit(lockin, image, tuple(klist[j]), sigma, workers)
                         for j in range(i, min(i + pool, len(klist)))}
            sf = ahead.pop(i).result()
        else:
            sf = lockin(image, (wx, wy), sigma, workers=workers)
        amp = np.abs(sf)
        take = amp > best_amp
        comp = carrier_1d(n0, -(wx - kref[0]))[:, None] * carrier_1d(n1, -(wy - kref[1]))[None, :]
        best = np.where(take, sf * comp, best)
        best_amp = np.where(take, amp, best_amp)
        kidx[take] = i
        if want_grad:
            ph = -np.angle(sf)
            g = np.stack(np.gradient(ph), axis=-1)
            g = g + TWO_PI * np.array([wx - kref[0], wy - kref[1]])
            grad = np.where(take[..., None], g, grad)
    if executor is not None:
        executor.shutdown()
    w = np.zeros((2, n0, n1))
    won = kidx >= 0
    w[0][won] = klist[kidx[won], 0]
    w[1][won] = klist[kidx[won], 1]
    out = {'lockin': best, 'kidx': kidx, 'w': w}
    if want_grad:
        out['grad'] = wrap_to_pi(2 * grad) / 2
    return out


def optwfr2(image, sigma, kx, ky, kw, kstep, workers=1):
    """gpa.py:669-686 signature; list built by :func:`sweep_grid`."""
    return sweep(image, sigma, sweep_grid(kx, ky, kw, kstep), (kx, ky), workers=workers)


def wfr2_grad_opt(image, sigma, kx, ky, kw, kstep, workers=1):
    """gpa.py:763-813 signature."""
    return sweep(image, sigma, sweep_grid(kx, ky, kw, kstep), (kx, ky),
                 want_grad=True, workers=workers)


def wfr4(image, sigma, klist, kref, dk):
    """Gated sweep over an ordered k-list: a candidate replaces the kept one only where its amplitude is
    strictly larger AND it lies within 2 sqrt(2) dk of the kept k-vector, which starts at klist[0] for
    every pixel.  Follows wfr4 (gpa.py:839-862).  Returns 'lockin', 'w' (2,N,M) and 'kidx' (-1 where
    nothing was ever accepted; 'w' is klist[0] there, as in the reference)."""
    image = np.asarray(image)
    klist = np.asarray(klist, dtype=np.float64).reshape(-1, 2)
    n0, n1 = image.shape
    best = np.zeros((n0, n1), dtype=np.complex128)
    w = np.zeros((n0, n1, 2))
    w[..., 0] = klist[0, 0]
    w[..., 1] = klist[0, 1]
    kidx = np.full((n0, n1), -1, dtype=np.int32)
    for i, (wx, wy) in enumerate(klist):
        sf = lockin(image, (wx, wy), sigma)
        sf = sf * (carrier_1d(n0, -(wx - kref[0]))[:, None] * carrier_1d(n1, -(wy - kref[1]))[None, :])
        t = np.abs(sf) > np.abs(best)
        t = t & (np.linalg.norm(w - np.array([wx, wy]), axis=-1) < 2 * np.sqrt(2) * dk)
        best[t] = sf[t]
        w[t] = np.array([wx, wy])
        kidx[t] = i
    return {'lockin': best, 'w': np.moveaxis(w, -1, 0), 'kidx': kidx}


def sweep_grad_variant(image, sigma, klist, kref, grad='diff', compensated=False):
    """The gradient-returning sweeps that are NOT wfr2_grad_opt:
      compensated=False: cuGPA.wfr2_grad_opt / wfr2_grad_single / wfr2_only_grad with grad='diff'
        (cu.py:58-66): forward differences of -angle(sf) along axis 0, then axis 1, NaN appended at the end
        of each axis, + 2 pi (w - kref), finally wrapToPi(2 g) / 2;
      compensated=True: wfr2_grad (gpa.py:722-760): the gradient function acts on the phase of the
        COMPENSATED lock-in and is wrapped per candidate; its 'diff' takes axis 1 first, then axis 0
        (gpa.py:738-742); grad=None is np.gradient.
    `grad` may also be a callable phase -> (N, M, 2) (compensated) or -> pair of (N, M) (cuGPA form)."""
    image = np.asarray(image)
    klist = np.asarray(klist, dtype=np.float64).reshape(-1, 2)
    n0, n1 = image.shape
    if grad == 'diff':
        if compensated:
            gf = lambda ph: np.stack([np.diff(ph, axis=1, append=np.nan), np.diff(ph, axis=0, append=np.nan)], axis=-1)
        else:
            gf = lambda ph: np.stack([np.diff(ph, axis=0, append=np.nan), np.diff(ph, axis=1, append=np.nan)], axis=-1)
    elif grad is None:
        gf = lambda ph: np.stack(np.gradient(ph), axis=-1)
    else:
        gf = grad if compensated else (lambda ph: np.stack(grad(ph), axis=-1))
    best = np.zeros((n0, n1), dtype=np.complex128)
    kidx = np.full((n0, n1), -1, dtype=np.int32)
    out_grad = np.zeros((n0, n1, 2))
    for i, (wx, wy) in enumerate(klist):
        sf = lockin(image, (wx, wy), sigma)
        comp = carrier_1d(n0, -(wx - kref[0]))[:, None] * carrier_1d(n1, -(wy - kref[1]))[None, :]
        t = np.abs(sf) > np.abs(best)
        if compensated:
            g = wrap_to_pi(gf(-np.angle(sf * comp)) * 2) / 2
        else:
            g = gf(-np.angle(sf)) + TWO_PI * np.array([wx - kref[0], wy - kref[1]])
        best = np.where(t, sf * comp, best)
        kidx[t] = i
        out_grad = np.where(t[..., None], g, out_grad)
    if not compensated:
        out_grad = wrap_to_pi(2 * out_grad) / 2
    w = np.zeros((2, n0, n1))
    won = kidx >= 0
    w[0][won] = klist[kidx[won], 0]
    w[1][won] = klist[kidx[won], 1]
    return {'lockin': best, 'kidx': kidx, 'w': w, 'grad': out_grad}


# --------------------------------------------------------------------------
# a5: phases / weights glue
# --------------------------------------------------------------------------
def derive_params(kvecs, sigma=None, kwscale=2.5, ksteps=3):
    """kw, sigma, kstep of extract_displacement_field (gpa.py:915-918)."""
    norms = np.linalg.norm(np.asarray(kvecs, dtype=np.float64), axis=1)
    kw = norms.mean() / kwscale
    if sigma is None:
        sigma = int(np.ceil(1 / norms.min()))
    return kw, sigma, kw / ksteps


def interior_mask(shape, dr):
    """Boolean mask, True on [dr:-dr, dr:-dr] (gpa.py:923-925)."""
    mask = np.zeros(shape, dtype=bool)
    mask[dr:-dr, dr:-dr] = True
    return mask


def phases_weights(lockins, sigma):
    """phases = angle(lockin); weights = |lockin| * (mask + 1e-6) (gpa.py:922-926)."""
    lockins = np.asarray(lockins)
    mask = interior_mask(lockins.shape[1:], 2 * sigma)
    return np.angle(lockins), np.abs(lockins) * (mask + 1e-6), mask


# --------------------------------------------------------------------------
# a6: per-pixel weighted least squares
# --------------------------------------------------------------------------
def weighted_lstsq(b, kmat, w):
    """Per pixel: argmin_x || w * (kmat @ x - b) ||, kmat (P,2), b/w (P,n,m').

    The reference calls LAPACK gelsd per pixel (myweighed_lstsq, gpa.py:97-113);
    this is the equivalent 2x2 normal-equation solve, with the min-norm answer
    at rank-deficient pixels (all-zero weights -> 0).  w is cropped to b's shape
    (weight of the left/top pixel of each difference, gpa.py:110).
    """
    b = np.asarray(b, dtype=np.float64)
    w = np.asarray(w, dtype=np.float64)[:, :b.shape[1], :b.shape[2]]
    ww = w * w
    k0 = kmat[:, 0][:, None, None]
    k1 = kmat[:, 1][:, None, None]
    a00 = (ww * k0 * k0).sum(0)
    a01 = (ww * k0 * k1).sum(0)
    a11 = (ww * k1 * k1).sum(0)
    r0 = (ww * k0 * b).sum(0)
    r1 = (ww * k1 * b).sum(0)
    det = a00 * a11 - a01 * a01
    tr = a00 + a11
    ok = det > 1e-28 * tr * tr
    safe = np.where(ok, det, 1.0)
    x0 = np.where(ok, (a11 * r0 - a01 * r1) / safe, 0.0)
    x1 = np.where(ok, (a00 * r1 - a01 * r0) / safe, 0.0)
    # rank-1 pixels: minimum-norm solution rhs / trace
    r1m = (~ok) & (tr > 0)
    safetr = np.where(tr > 0, tr, 1.0)
    x0 = np.where(r1m, r0 / safetr, x0)
    x1 = np.where(r1m, r1 / safetr, x1)
    return np.stack([x0, x1])


def reconstruct_gradients(kvecs, phases, weights):
    """Wrapped phase differences -> displacement-gradient fields.

    dbdx = wrap(diff(phases, axis=2)), dbdy = wrap(diff(phases, axis=1))
    (gpa.py:234-235), each solved per pixel against K = 2 pi kvecs
    (gpa.py:227, :236-237).  Returns dudx (2,N,M-1), dudy (2,N-1,M).
    """
    kmat = TWO_PI * np.asarray(kvecs, dtype=np.float64)
    dbdx = wrap_to_pi(np.diff(phases, axis=2))
    dbdy = wrap_to_pi(np.diff(phases, axis=1))
    return weighted_lstsq(dbdx, kmat, weights), weighted_lstsq(dbdy, kmat, weights)


# --------------------------------------------------------------------------
# a7: DCT-Laplacian weighted phase unwrap (Ghiglia-Romero PCG)
# --------------------------------------------------------------------------
def poisson_scale(shape, compat=True):
    """Eigenvalues of the Neumann Laplacian used as DCT-domain divisor.

    compat=True replicates precomp_Poissonscaling (pu.py:106-115) literally,
    including its swapped axes 2(cos(pi I/M)+cos(pi J/N)-2) for I<N, J<M
    (identical for square images); compat=False uses the proper cos(pi I/N)+
    cos(pi J/M).  [0,0] -> 1.
    """
    n, m = shape
    i = np.arange(n)[:, None]
    j = np.arange(m)[None, :]
    if compat:
        scale = 2 * (np.cos(np.pi * i / m) + np.cos(np.pi * j / n) - 2)
    else:
        scale = 2 * (np.cos(np.pi * i / n) + np.cos(np.pi * j / m) - 2)
    scale[0, 0] = 1.0
    return scale


def apply_q(p, wwx, wwy):
    """A^T diag(WWx,WWy) A p with zero-padded outer differences (pu.py:118-132)."""
    fx = wwx * (p[:, 1:] - p[:, :-1])
    fy = wwy * (p[1:, :] - p[:-1, :])
    q = np.zeros_like(p)
    q[:, :-1] += fx
    q[:, 1:] -= fx
    q[:-1, :] += fy
    q[1:, :] -= fy
    return q


def unwrap_prediff(dx, dy, weight=None, kmax=100, eps=1e-9, compat=True,
                   workers=1, return_iters=False):
    """Weighted least-squares unwrap from pre-differenced gradients.

    dx (N,M-1) = differences along axis 1, dy (N-1,M) along axis 0, both
    re-wrapped to [-pi,pi) first (pu.py:296-297).  Edge weights = min of the
    squared weights of the two pixels (pu.py:305-310).  PCG with the DCT Poisson
    solve as preconditioner, in the operation order of pu.py:326-349.
    Follows phase_unwrap_prediff (pu.py:282-350).
    """
    dx = wrap_to_pi(np.asarray(dx, dtype=np.float64))
    dy = wrap_to_pi(np.asarray(dy, dtype=np.float64))
    n, m = dx.shape[0], dy.shape[1]
    if weight is None:
        wwx = np.ones_like(dx)
        wwy = np.ones_like(dy)
    else:
        ww = np.asarray(weight, dtype=np.float64) ** 2
        wwx = np.minimum(ww[:, :-1], ww[:, 1:])
        wwy = np.minimum(ww[:-1, :], ww[1:, :])
    fx = wwx * dx
    fy = wwy * dy
    r = np.zeros((n, m))
    r[:, :-1] += fx
    r[:, 1:] -= fx
    r[:-1, :] += fy
    r[1:, :] -= fy
    # r[j] = f[j] - f[j-1] with zero-padded ends: np.diff(prepend=0, append=0)
    # of pu.py:315-316.
    norm0 = np.linalg.norm(r)
    phi = np.zeros((n, m))
    scale = poisson_scale((n, m), compat=compat)
    k = 0
    rho_prev = None
    p = None
    while np.any(r != 0.0):
        z = sfft.idctn(sfft.dctn(r, workers=workers) / scale, workers=workers)
        k += 1
        rho = np.vdot(r, z)
        p = z if k == 1 else z + (rho / rho_prev) * p
        rho_prev = rho
        q = apply_q(p, wwx, wwy)
        alpha = rho / np.vdot(p, q)
        phi = phi + alpha * p
        r = r - alpha * q
        if k >= kmax or np.linalg.norm(r) < eps * norm0:
            break
    return (phi, k) if return_iters else phi


def unwrap(psi, weight=None, kmax=100, **kw):
    """Unwrap a wrapped phase image: differences wrapped, then as unwrap_prediff.

    Follows phase_unwrap (pu.py:141-208).
    """
    psi = np.asarray(psi, dtype=np.float64)
    return unwrap_prediff(np.diff(psi, axis=1), np.diff(psi, axis=0),
                          weight=weight, kmax=kmax, **kw)


# --------------------------------------------------------------------------
# a5+a6+a7 driver pieces and the top-level entry point
# --------------------------------------------------------------------------
def reconstruct_u_inv_from_phases(kvecs, phases, weights, weighted_unwrap=True,
                                  kmax=10, workers=1, return_iters=False, pre_diff=False):
    """gpa.py:196-245.  pre_diff=True (gpa.py:228-232): `phases` is (P, N, M, 2) and already holds the phase
    gradients along axis 1 ([..., 0]) and axis 0 ([..., 1]); they are wrapped and cropped to the
    difference grids instead of being differenced."""
    if pre_diff:
        kmat = TWO_PI * np.asarray(kvecs, dtype=np.float64)
        phases = np.asarray(phases, dtype=np.float64)
        dudx = weighted_lstsq(wrap_to_pi(phases[..., 0])[:, :, :-1], kmat, weights)
        dudy = weighted_lstsq(wrap_to_pi(phases[..., 1])[:, :-1], kmat, weights)
    else:
        dudx, dudy = reconstruct_gradients(kvecs, phases, weights)
    wn = np.linalg.norm(weights, axis=0) if weighted_unwrap else None
    us, iters = [], []
    for i in range(2):
        if weighted_unwrap:
            phi, it = unwrap_prediff(dudx[i], dudy[i], wn, kmax=kmax,
                                     workers=workers, return_iters=True)
        else:
            phi, it = unwrap_prediff(dudx[i], dudy[i], workers=workers, return_iters=True)
        us.append(phi)
        iters.append(it)
    u = np.array(us)
    return (u, iters) if return_iters else u


def extract_displacement_field(image, kvecs, sigma=None, kwscale=2.5, ksteps=3,
                               klists=None, workers=1, return_parts=False, pool=1):
    """Top-level path, gpa.py:907-932 (deconvolve=False).

    ``klists``: optional list of P explicit (K,2) k-lists (wfr3-style,
    gpa.py:647-666) replacing the np.arange grid -- used for the BASELINE
    configs whose K is not a square number.
    """
    image = np.asarray(image, dtype=np.float64)
    kvecs = np.asarray(kvecs, dtype=np.float64)
    kw, sigma, kstep = derive_params(kvecs, sigma, kwscale, ksteps)
    img0 = image - image.mean()
    gs = []
    for p, pk in enumerate(kvecs):
        kl = sweep_grid(pk[0], pk[1], kw, kstep) if klists is None else klists[p]
        gs.append(sweep(img0, sigma, kl, pk, workers=1 if pool > 1 else workers, pool=pool))
    lockins = np.stack([g['lockin'] for g in gs])
    phases, weights, _ = phases_weights(lockins, sigma)
    u, iters = reconstruct_u_inv_from_phases(kvecs, phases, weights, workers=workers,
                                             return_iters=True)
    if return_parts:
        return u, {'gs': gs, 'phases': phases, 'weights': weights, 'iters': iters,
                   'sigma': sigma, 'kw': kw, 'kstep': kstep}
    return u


# --------------------------------------------------------------------------
# a8: single-reference-vector route
# --------------------------------------------------------------------------
def fit_plane(image):
    """Huber-loss plane fit a0*x + a1*y + a2 (mt.py:30-47)."""
    xx, yy = np.meshgrid(np.arange(image.shape[0]), np.arange(image.shape[1]), indexing='ij')

    def resid(a):
        return (image - (a[0] * xx + a[1] * yy + a[2])).ravel()
    return spo.least_squares(resid, np.zeros(3), loss='huber').x


def iterate_gpa(image, kvecs, sigma, edge=5, iters=3, kmax_iter=25, kmax=200):
    """Reference-vector refinement loop, gpa.py:116-154 (+ fit_delta_k :92-94)."""
    kvecs = np.asarray(kvecs, dtype=np.float64)
    corr = np.zeros_like(kvecs)
    for i in range(iters + 1):
        rs = lockin_batch(image, kvecs + corr, sigma)
        sl = (slice(edge, -edge), slice(edge, -edge)) if edge > 0 else (slice(None), slice(None))
        prs = [np.angle(r)[sl] for r in rs]
        w = np.stack([np.abs(r)[sl] for r in rs])
        last = i == iters
        prs = [unwrap(pr, np.sqrt(we / we.max()), kmax=kmax if last else kmax_iter)
               for pr, we in zip(prs, w)]
        if not last:
            corr = corr - np.stack([fit_plane(pr)[:2] / TWO_PI for pr in prs])
    return np.stack(prs), w, corr


def reconstruct_u_inv(kvecs, b, weights=None):
    """Unwrapped phases -> u (gpa.py:157-193, use_only_ks=None branches)."""
    kmat = TWO_PI * np.asarray(kvecs, dtype=np.float64)
    b = b - b.mean(axis=(1, 2), keepdims=True)
    if weights is None:
        sol = np.linalg.lstsq(kmat, b.reshape((b.shape[0], -1)), rcond=None)[0]
        return sol.reshape((2,) + b.shape[1:])
    return weighted_lstsq(b, kmat, weights)


# --------------------------------------------------------------------------
# f-1: Lawler-Fujita undistortion
# --------------------------------------------------------------------------
class _Resampler:
    """scipy.ndimage.map_coordinates(field, coords, order=3, mode=mode[, cval]) for MANY coordinate sets of ONE field: the
    cubic-spline prefilter of the field runs once instead of once per call.  map_coordinates itself is
    `padded, npad = _prepad_for_spline_filter(input, mode, cval); filtered = spline_filter(padded, 3, float64, mode);
    _nd_image.geometric_transform(filtered, None, coords, None, None, output, 3, mode_code, cval, npad, None, None)`
    (scipy/ndimage/_interpolation.py); this class keeps `filtered` and repeats the last call -- the same C routine with the
    same arguments, so the numbers are map_coordinates' bit for bit (tests/test_oracle_golden.py holds it to that).  If
    SciPy's private layout ever changes, `fast` is False and every sample is a plain map_coordinates call."""

    def __init__(self, field, mode):
        import scipy.ndimage as ndi
        self.field = np.asarray(field, dtype=np.float64)
        self.mode = mode
        self.fast = False
        try:
            from scipy.ndimage import _interpolation as _ip, _nd_image, _ni_support
            padded, self.npad = _ip._prepad_for_spline_filter(self.field, mode, 0.0)
            self.filtered = ndi.spline_filter(padded, 3, output=np.float64, mode=mode)
            self.code = _ni_support._extend_mode_to_code(mode)
            self._gt = _nd_image.geometric_transform
            self.fast = True
        except Exception:
            pass

    def __call__(self, coords, cval=0.0):
        import scipy.ndimage as ndi
        coords = np.asarray(coords, dtype=np.float64)
        if not self.fast:
            return ndi.map_coordinates(self.field, coords, mode=self.mode, cval=cval)
        out = np.zeros(coords.shape[1:], dtype=np.float64)
        self._gt(self.filtered, None, coords, None, None, out, 3, self.code, cval, self.npad, None, None)
        return out


def invert_u_overlap(us, iters=35, edge=0, mode='nearest', rows=None):
    """Fixed-point inversion of a displacement field: u_it(r) <- u(r + u_it(r)), `iters` rounds of
    cubic-spline resampling (scipy.ndimage.map_coordinates, order 3) on the grid extended by
    `edge` pixels; the last round is called with cval=nan.  Follows invert_u_overlap
    (gpa.py:262-300).
    rows (test aid for large fields): a 1-D array of row indices of the output grid; the iterate of a pixel depends on the
    field alone, never on the other pixels' iterates, so those rows come out exactly as in the full call (result: (2, len(rows), M))."""
    us = np.asarray(us, dtype=np.float64)
    xx, yy = np.mgrid[-edge:us.shape[1] + edge, -edge:us.shape[2] + edge]
    if rows is not None:
        xx, yy = xx[np.asarray(rows)], yy[np.asarray(rows)]
    sample = [_Resampler(u, mode) for u in us]
    u_it = [f([xx, yy]) for f in sample]
    for _ in range(iters - 1):
        u_it = [f([xx + u_it[0], yy + u_it[1]]) for f in sample]
    u_it = [f([xx + u_it[0], yy + u_it[1]], cval=np.nan) for f in sample]
    return np.stack(u_it)


def invert_u(us, iters=35, edge=0, mode='nearest'):
    """The variant without overlap (gpa.py:248-259): the grid is the image's own, every one of the
    `iters` rounds samples at r + u_it(r) - edge."""
    import scipy.ndimage as ndi
    us = np.asarray(us, dtype=np.float64)
    xx, yy = np.mgrid[:us.shape[1], :us.shape[2]]
    u_it = [ndi.map_coordinates(u, [xx, yy], mode=mode) for u in us]
    for _ in range(iters):
        u_it = [ndi.map_coordinates(u, [xx + u_it[0] - edge, yy + u_it[1] - edge], mode=mode) for u in us]
    return np.stack(u_it)


def undistort_image(deformed, u, rows=None):
    """Resample `deformed` at r + u_inv(r), u_inv = invert_u_overlap(-u); the resampling uses
    map_coordinates' defaults (order 3, mode='constant', cval=0).  Follows undistort_image
    (gpa.py:935-974).  rows: as in invert_u_overlap (those rows of the result)."""
    import scipy.ndimage as ndi
    u = np.asarray(u, dtype=np.float64)
    u_inv = invert_u_overlap(-u, rows=rows)
    xx, yy = np.mgrid[:u.shape[1], :u.shape[2]]
    if rows is not None:
        xx, yy = xx[np.asarray(rows)], yy[np.asarray(rows)]
    return ndi.map_coordinates(np.asarray(deformed, dtype=np.float64), [xx + u_inv[0], yy + u_inv[1]])


# --------------------------------------------------------------------------
# f-2: phase gradient -> Jacobian -> lattice properties
# --------------------------------------------------------------------------
def calc_diff_from_isotropic(ani_ks, symmetry=6):
    """dks such that ani_ks + dks is an isotropic lattice of the mean radius at the periodic-mean
    angle (geometric_phase_analysis.py:303-322)."""
    ani_ks = np.asarray(ani_ks, dtype=np.float64)
    period = TWO_PI / symmetry
    ang = np.arctan2(ani_ks[:, 1], ani_ks[:, 0])
    dt = np.angle(np.exp(1j * TWO_PI / period * ang).mean()) * period / TWO_PI
    r = np.linalg.norm(ani_ks, axis=1).mean()
    th = dt + period * np.arange(symmetry)
    ks_hex = r * np.stack([np.cos(th), np.sin(th)], axis=-1)
    alldiffs = ks_hex - ani_ks[:, None]
    argmins = np.linalg.norm(alldiffs, axis=-1).argmin(axis=1)
    return alldiffs[np.arange(len(ani_ks)), argmins]


def phasegradient2J(kvecs, grads, weights, nmperpixel, iso_ref=False):
    """J (N,M,2,2) from the sweep's phase gradients (P,N,M,2) by the per-pixel weighted least
    squares (property_extract.py:69-101, sort=0)."""
    kvecs = np.asarray(kvecs, dtype=np.float64)
    grads = np.asarray(grads, dtype=np.float64)
    if iso_ref:
        dks = calc_diff_from_isotropic(kvecs)
        kmat = TWO_PI * (kvecs + dks)
        grads = wrap_to_pi(grads - TWO_PI * dks[:, None, None, :])
    else:
        kmat = TWO_PI * kvecs
    dudx = weighted_lstsq(grads[..., 0], kmat, weights)
    dudy = weighted_lstsq(grads[..., 1], kmat, weights)
    J = np.stack([dudx, dudy], axis=-1) / nmperpixel
    return np.moveaxis(J, 0, -2)


def props_from_jac(jac, refangle=0., refscale=1., diff=False):
    """(angle, aniangle, alpha, kappa) of a lattice from the Jacobian of its transformation, by
    the sign-normalised SVD of property_extract.py:137-178."""
    u, s, v = np.linalg.svd(np.asarray(jac, dtype=np.float64))
    signs = np.sign(u[..., None, [0, 1], [0, 1]])
    v = signs * v
    u = np.swapaxes(signs * u, -1, -2)
    u_p = np.swapaxes(u @ v, -1, -2)
    angle = np.rad2deg(np.arctan2(u_p[..., 1, 0], u_p[..., 0, 0]))
    aniangle = np.rad2deg(np.arctan2(u[..., 1, 0], u[..., 0, 0]))
    if diff:
        aniangle = aniangle + 90
        alpha = s[..., 0]
    else:
        alpha = s[..., 1]
    kappa = s[..., 0] / s[..., 1]
    return np.array([angle + refangle, aniangle % 180, alpha * refscale, kappa])


# --------------------------------------------------------------------------
# a9: smooth + periodic decomposition (PARITY UNPINNED, see module docstring)
# --------------------------------------------------------------------------
def per(image, inverse_dft=True):
    """Moisan (2011) periodic + smooth decomposition.

    Call site gpa.py:429 uses per(image, inverse_dft=False)[0] = DFT of the
    periodic component.  s_hat = v_hat / (2cos(2 pi q/N) + 2cos(2 pi r/M) - 4),
    s_hat[0,0] = 0, v = border-jump image; p_hat = u_hat - s_hat.
    """
    u = np.asarray(image, dtype=np.float64)
    n, m = u.shape
    v = np.zeros_like(u)
    d0 = u[-1, :] - u[0, :]
    v[0, :] += d0
    v[-1, :] -= d0
    d1 = u[:, -1] - u[:, 0]
    v[:, 0] += d1
    v[:, -1] -= d1
    vhat = np.fft.fft2(v)
    q = np.arange(n)[:, None]
    r = np.arange(m)[None, :]
    den = 2 * np.cos(TWO_PI * q / n) + 2 * np.cos(TWO_PI * r / m) - 4
    den[0, 0] = 1.0
    shat = vhat / den
    shat[0, 0] = 0.0
    phat = np.fft.fft2(u) - shat
    if inverse_dft:
        return np.real(np.fft.ifft2(phat)), np.real(np.fft.ifft2(shat))
    return phat, shat


# --------------------------------------------------------------------------
# f-3: peak finding -- extract_primary_ks, gpa.py:397-548.
# Pinned against the reference's own driver logic with two declared stand-ins for third-party
# functions that are absent from this image (oracle/make_golden.py): moisan2011.per -> per() above,
# skimage.feature.peak_local_max -> peak_local_max() below (restated from scikit-image >= 0.19).
# --------------------------------------------------------------------------
def peak_local_max(image, threshold_rel):
    """skimage.feature.peak_local_max(image, threshold_rel=...) with its defaults min_distance=1,
    exclude_border=True: pixels equal to the maximum of their 3x3 neighbourhood and above
    max(image.min(), threshold_rel * image.max()), one border pixel excluded, highest first
    (stable, i.e. raster order among equals); ensure_spacing is a no-op at min_distance 1."""
    import scipy.ndimage as ndi
    image = np.asarray(image)
    thr = max(image.min(), threshold_rel * image.max())
    mask = (image == ndi.maximum_filter(image, footprint=np.ones((3, 3)), mode='nearest'))
    if np.all(mask):
        mask[:] = False
    mask &= image > thr
    mask[0, :] = mask[-1, :] = False
    mask[:, 0] = mask[:, -1] = False
    coord = np.nonzero(mask)
    order = np.argsort(-image[coord], kind='stable')
    return np.transpose(coord)[order]


def smoothed_spectrum(image, sigma=1, DoG=True):
    """|fftshift(per(image - mean))| smoothed by a Gaussian (minus a sigma=50 one), gpa.py:427-434."""
    import scipy.ndimage as ndi
    image = np.asarray(image, dtype=np.float64)
    pd, _ = per(image - image.mean(), inverse_dft=False)
    fftim = np.abs(np.fft.fftshift(pd))
    smooth = ndi.gaussian_filter(fftim, sigma=sigma)
    if DoG:
        smooth -= ndi.gaussian_filter(fftim, sigma=50)
    return smooth


def fftbounds(n, d=1):
    """imagetools.py:22-26"""
    r = np.fft.fftshift(np.fft.fftfreq(n, d))
    return np.append(r, r[-1] + 1 / (n * d))


def remove_negative_duplicates(ks):
    """mathtools.py:78-94"""
    if ks.shape[0] == 0:
        return ks
    nonneg = np.where(np.sign(ks[:, [0]]) != 0, np.sign(ks[:, [0]]) * ks, np.sign(ks[:, [1]]) * ks)
    npks = [nonneg[0]]
    atol = 1e-3 * np.min(np.abs(nonneg), axis=1).mean()
    for k in nonneg[1:]:
        if not np.any(np.all(np.isclose(k, npks, atol=atol), axis=1)):
            npks.append(k)
    return np.array(npks)


def _decrease_threshold(t):
    """gpa.py:388-394"""
    if t > 0.001:
        t = t - 0.1 if t >= 0.2 else t / 2
    return t


def smallest_sum(ks):
    """gpa.py:538-548"""
    M = np.ones((3, 3)) - 2 * np.eye(3)
    sums = M @ ks
    return sums[np.argmin(np.linalg.norm(sums, axis=1))]


def select_closest_to_triangle(ks):
    """gpa.py:529-535"""
    from itertools import combinations
    combis = list(combinations(ks, 3))
    sums = [np.linalg.norm(smallest_sum(np.array(c))) for c in combis]
    return np.array(combis[int(np.argmin(sums))])


def extract_primary_ks(image, threshold=0.7, pix_norm_range=(2, 200), sigma=1, DoG=True):
    """gpa.py:397-505 (plotting dropped).  As in the reference, the recursive calls do not pass DoG on
    (it falls back to its default True)."""
    image = np.asarray(image, dtype=np.float64)
    smooth = smoothed_spectrum(image, sigma, DoG)
    kxs, kys = [fftbounds(n) for n in smooth.shape]
    center = np.array(smooth.shape) // 2
    cindices = peak_local_max(smooth, threshold_rel=threshold)
    coords = cindices - center
    norms = np.linalg.norm(coords, axis=1)
    selection = np.logical_and(norms < pix_norm_range[1], norms > pix_norm_range[0])
    cindices = cindices[selection]
    coords = coords[selection]
    all_ks = np.array([kxs[cindices.T[0]], kys[cindices.T[1]]]).T
    all_ks = remove_negative_duplicates(all_ks)
    newparams = False
    if len(all_ks) < 3:
        newparams = True
        if len(all_ks) == 0:
            if threshold > _decrease_threshold(threshold):
                threshold = _decrease_threshold(threshold)
            else:
                newparams = False
        else:
            coordsminlength = np.linalg.norm(coords, axis=1).min()
            top = 0.2 * np.max([smooth[c[0], c[1]] for c in cindices])
            if coordsminlength < 5 * sigma:
                sigma = coordsminlength / 6
            elif threshold > top:
                threshold = top
            elif threshold > _decrease_threshold(threshold):
                threshold = _decrease_threshold(threshold)
            else:
                newparams = False
        if newparams:
            primary_ks, all_ks = extract_primary_ks(image, threshold=threshold, sigma=sigma, pix_norm_range=pix_norm_range)
        else:
            primary_ks = all_ks.copy()
    if not newparams:
        primary_ks = all_ks.copy()
    if len(primary_ks) != 3:
        if len(primary_ks) > 3:
            primary_ks = select_closest_to_triangle(all_ks)
        elif len(all_ks) > 6:
            primary_ks = select_closest_to_triangle(all_ks)
        elif threshold > _decrease_threshold(threshold) and not newparams:
            threshold = _decrease_threshold(threshold)
            primary_ks, all_ks = extract_primary_ks(image, threshold=threshold, sigma=sigma, pix_norm_range=pix_norm_range)
        else:
            primary_ks = all_ks.copy()
    return primary_ks, all_ks


# --------------------------------------------------------------------------
# f-4 (second half): gaussian_deconvolve, gpa.py:892-904, over skimage.restoration.wiener
# (absent from this image; restated from scikit-image's restoration/deconvolution.py + uft.py).
# --------------------------------------------------------------------------
def wiener(image, psf, balance):
    """skimage.restoration.wiener(image, psf, balance, clip=False, is_real=True) for a real-space psf:
    irfft2(conj(H) / (|H|^2 + balance |L|^2) * rfft2(image)), H / L the transfer functions (uft.ir2tf)
    of the psf and of the 3x3 Laplacian stencil, both centred on the origin."""
    image = np.asarray(image, dtype=np.float64)

    def ir2tf(imp, shape):
        pad = np.zeros(shape)
        pad[tuple(slice(0, s) for s in imp.shape)] = imp
        for axis, size in enumerate(imp.shape):
            pad = np.roll(pad, shift=-int(np.floor(size / 2)), axis=axis)
        return np.fft.rfft2(pad)
    lap = np.zeros((3, 3))
    lap[1, :] = -1.0
    lap[:, 1] = -1.0
    lap[1, 1] = 4.0
    reg = ir2tf(lap, image.shape)
    H = ir2tf(np.asarray(psf).real, image.shape)
    filt = np.conj(H) / (np.abs(H) ** 2 + balance * np.abs(reg) ** 2)
    return np.fft.irfft2(filt * np.fft.rfft2(image), s=image.shape)


def gaussian_deconvolve(data, sigma, dr=20, balance=5000):
    """gpa.py:892-904: reflect-pad by 2 dr, Wiener-deconvolve with the real-space image of the k-space
    Gaussian, crop."""
    import scipy.ndimage as ndi
    data = np.asarray(data, dtype=np.float64)
    padding = [(0, 0)] * (data.ndim - 2) + [(2 * dr, 2 * dr), (2 * dr, 2 * dr)]
    padded = np.pad(data, padding, mode='reflect')
    kernel = np.fft.fft2(ndi.fourier_gaussian(np.ones(padded.shape[-2:]), sigma=sigma)).real
    kernel = np.fft.fftshift(kernel)
    kernel = kernel / kernel.sum()
    dec = [wiener(p, kernel, balance)[2 * dr:-2 * dr, 2 * dr:-2 * dr] for p in padded.reshape((-1,) + padded.shape[-2:])]
    return np.reshape(np.stack(dec), data.shape)
