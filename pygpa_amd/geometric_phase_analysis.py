"""Host-side mirror of the hot path of pyGPA/geometric_phase_analysis.py.

Same function names, argument order, defaults and return containers as the
reference; the arithmetic runs in libgpa_hip.so on an MI355X through the C ABI of
include/gpa_hip.h.  NumPy arrays in, fresh NumPy arrays out.  There is no CPU
fallback: without the HIP library or a GPU every function raises GPAError.

Extra keyword ``dtype`` (np.float64 default, like the reference's complex128
path; np.float32 for the single-precision kernels) is the only addition.
"""
import numpy as np

from . import _lib
from ._lib import GPAError  # noqa: F401
from .mathtools import wrapToPi, fit_plane, periodic_average, remove_negative_duplicates  # noqa: F401

DEFAULT_DTYPE = np.float64


def _plan(image, batch, dtype):
    image = np.asarray(image)
    if image.ndim != 2:
        raise ValueError('image must be 2-D')
    return _lib.get_plan(image.shape, batch, DEFAULT_DTYPE if dtype is None else dtype)


def _sweep_list(kx, ky, kw, kstep):
    """k-list of the reference's double loop, wx outer / wy inner.  Built with
    np.arange on the host because its length depends on float rounding
    (geometric_phase_analysis.py:679-680)."""
    wxs = np.arange(kx - kw, kx + kw, kstep)
    wys = np.arange(ky - kw, ky + kw, kstep)
    return np.array([(wx, wy) for wx in wxs for wy in wys], dtype=np.float64).reshape(-1, 2)


def _w_from_kidx(kidx, klist):
    w = np.zeros((2,) + kidx.shape)
    won = kidx >= 0
    w[0][won] = klist[kidx[won], 0]
    w[1][won] = klist[kidx[won], 1]
    return w


# --------------------------------------------------------------------------- a1 / a2
def GPA(image, kx, ky, sigma=22, dtype=None):
    """Spatial lock-in (geometric_phase_analysis.py:20-45)."""
    return _plan(image, 1, dtype).lockin_batch(image, [(kx, ky)], sigma)[0]


def optGPA(image, kvec, sigma=22, dtype=None):
    """Spatial lock-in, k-vector as a pair (geometric_phase_analysis.py:48-76)."""
    return _plan(image, 1, dtype).lockin_batch(image, [tuple(kvec)], sigma)[0]


def vecGPA(image, kvecs, sigma=22, dtype=None):
    """Batched lock-in over a (K,2) list (geometric_phase_analysis.py:79-89)."""
    kvecs = np.asarray(kvecs, dtype=np.float64).reshape(-1, 2)
    return _plan(image, len(kvecs), dtype).lockin_batch(image, kvecs, sigma)


# --------------------------------------------------------------------------- a3 / a4
def wfr3(image, sigma, klist, kref, dtype=None):
    """Adaptive lock-in over an explicit k-list (geometric_phase_analysis.py:647-666)."""
    klist = np.asarray(klist, dtype=np.float64).reshape(-1, 2)
    lockin, kidx, _ = _plan(image, len(klist), dtype).sweep(image, kref, klist, sigma)
    return {'w': _w_from_kidx(kidx, klist), 'lockin': lockin, 'kidx': kidx}


def optwfr2(image, sigma, kx, ky, kw, kstep, dtype=None):
    """Adaptive lock-in on the np.arange grid around (kx, ky)
    (geometric_phase_analysis.py:669-686); identical results to wfr2 (:615-644)."""
    return wfr3(image, sigma, _sweep_list(kx, ky, kw, kstep), (kx, ky), dtype=dtype)


wfr2 = optwfr2


def wfr2_only_lockin(image, sigma, kx, ky, kw, kstep, dtype=None):
    """Only the lock-in array (geometric_phase_analysis.py:689-702)."""
    klist = _sweep_list(kx, ky, kw, kstep)
    return _plan(image, len(klist), dtype).sweep(image, (kx, ky), klist, sigma, want_kidx=False)[0]


def _grad_sweep(image, sigma, klist, kref, grad, compensated, dtype):
    """Sweep + phase gradient of the winner for every gradient spelling of the reference.
    grad None: np.gradient; 'diff': forward differences (NaN at the end of each axis), in the component order of
    the module that is being mirrored; a callable: the lock-ins of all candidates come from the device
    (gpa_lockin_batch) and the callable -- host Python by its nature -- is applied per candidate as the
    reference does."""
    klist = np.asarray(klist, dtype=np.float64).reshape(-1, 2)
    plan = _plan(image, len(klist), dtype)
    if grad is None or grad == 'diff':
        mode = 0 if grad is None else (2 if compensated else 1)
        lockin, kidx, g = plan.sweep(image, kref, klist, sigma, want_grad=True, grad_mode=mode)
        return {'w': _w_from_kidx(kidx, klist), 'lockin': lockin, 'grad': g, 'kidx': kidx}
    if not callable(grad):
        raise ValueError("grad must be None, 'diff' or a callable")
    image = np.asarray(image)
    n0, n1 = image.shape
    best = np.zeros((n0, n1), dtype=plan.cdtype)
    kidx = np.full((n0, n1), -1, dtype=np.int32)
    out = np.zeros((n0, n1, 2), dtype=plan.rdtype)
    xx, yy = np.ogrid[0:n0, 0:n1]
    for start in range(0, len(klist), 8):
        sfs = plan.lockin_batch(image, klist[start:start + 8], sigma)
        for i, sf in enumerate(sfs, start):
            wx, wy = klist[i]
            comp = np.exp(-2j * np.pi * ((wx - kref[0]) * xx + (wy - kref[1]) * yy))
            t = np.abs(sf) > np.abs(best)
            if compensated:      # wfr2_grad (geometric_phase_analysis.py:750-756)
                g = wrapToPi(np.asarray(grad(-np.angle(sf * comp))) * 2) / 2
            else:                # cuGPA (cuGPA.py:75-80)
                g = np.stack(grad(-np.angle(sf)), axis=-1) + 2 * np.pi * np.array([wx - kref[0], wy - kref[1]])
            best = np.where(t, sf * comp, best)
            kidx[t] = i
            out = np.where(t[..., None], g, out)
    if not compensated:
        out = wrapToPi(2 * out) / 2
    return {'w': _w_from_kidx(kidx, klist), 'lockin': best, 'grad': out, 'kidx': kidx}


def wfr2_grad_opt(image, sigma, kx, ky, kw, kstep, dtype=None):
    """Adaptive lock-in that also returns the winner's phase gradient
    (geometric_phase_analysis.py:763-813)."""
    return _grad_sweep(image, sigma, _sweep_list(kx, ky, kw, kstep), (kx, ky), None, False, dtype)


def wfr2_grad(image, sigma, kx, ky, kw, kstep, grad=None, dtype=None):
    """geometric_phase_analysis.py:722-760: the gradient function acts on the phase of the compensated lock-in
    (grad=None np.gradient -- the numbers of wfr2_grad_opt up to rounding --, 'diff' forward differences with
    [..., 0] along axis 1 and [..., 1] along axis 0 (:738-742), or a callable phase -> (N, M, 2))."""
    return _grad_sweep(image, sigma, _sweep_list(kx, ky, kw, kstep), (kx, ky), grad, True, dtype)


def wfr2_grad_vec(image, sigma, kx, ky, kw, kstep, dtype=None):
    """The dask-vectorised spelling of wfr2_grad_opt (geometric_phase_analysis.py:816-836): same numbers; here
    the candidates are the batch dimension of the device kernels anyway."""
    return wfr2_grad_opt(image, sigma, kx, ky, kw, kstep, dtype=dtype)


def wfr2_only_lockin_vec(image, sigma, kx, ky, kw, kstep, dtype=None):
    """The dask-vectorised spelling of wfr2_only_lockin (geometric_phase_analysis.py:705-719)."""
    return wfr2_only_lockin(image, sigma, kx, ky, kw, kstep, dtype=dtype)


def wfr4(image, sigma, klist, kref, dk, dtype=None):
    """Sweep over an ORDERED k-list in which a candidate is accepted only where its amplitude is larger and it
    lies within 2 sqrt(2) dk of the k-vector kept so far (geometric_phase_analysis.py:839-862).  The distance
    test depends only on (kept candidate, new candidate): it is evaluated here in double for all K x K pairs and
    the device kernel looks it up (gpa_sweep_gated)."""
    klist = np.asarray(klist, dtype=np.float64).reshape(-1, 2)
    gate = np.linalg.norm(klist[:, None, :] - klist[None, :, :], axis=-1) < 2 * np.sqrt(2) * dk
    lockin, kidx = _plan(image, len(klist), dtype).sweep_gated(image, kref, klist, sigma, gate)
    w = _w_from_kidx(kidx, klist)
    never = kidx < 0
    w[0][never], w[1][never] = klist[0]          # the reference starts every pixel at klist[0] (:850-851)
    return {'w': w, 'lockin': lockin, 'kidx': kidx}


def generate_klists(pks, dk=None, kmax=1.9, kmin=0.2, sort_list=False):
    """Candidate lists for wfr3 / wfr4 (geometric_phase_analysis.py:865-889).  A fixed 0.005-spaced k-grid covers the
    square of half-width kmax * max|k|; a grid point goes to peak i when it lies in the ring kmin .. kmax (times
    max|k|) and no peak or negated peak is strictly closer to it than peak i (so a point equidistant from two peaks
    is in both lists).  Points come out in row-major grid order, or sorted by distance from the peak.

    Built as a running nearest-site search over the 2 P sites on one (n x n) plane at a time -- the Voronoi cell of
    each peak among {+k, -k} -- rather than a (n, n, 2 P) distance cube; `dk` is accepted and unused like in the
    reference.  Host bookkeeping, pinned by tests/golden/variants_64.npz ('wfr4_ring') and tests/test_host_logic.py."""
    sites = np.asarray(pks, dtype=np.float64).reshape(-1, 2)
    npk = len(sites)
    sites = np.vstack([sites, -sites])
    scale = np.sqrt((sites[:npk] ** 2).sum(axis=1)).max()
    outer, inner = scale * kmax, scale * kmin
    step = 0.005
    # the axis of np.mgrid[-outer:outer:step]: start + i * step
    axis = np.arange(int(np.ceil((outer - (-outer)) / step))) * step + (-outer)
    gx, gy = axis[:, None], axis[None, :]
    rr = gx ** 2 + gy ** 2
    keep = (rr < outer ** 2) & (rr > inner ** 2)

    def dist2(site):
        return (gx - site[0]) ** 2 + (gy - site[1]) ** 2

    nearest = dist2(sites[0])
    for site in sites[1:]:
        np.minimum(nearest, dist2(site), out=nearest)
    lists = []
    for i in range(npk):
        ix, iy = np.nonzero(keep & (dist2(sites[i]) == nearest))
        pts = np.column_stack([axis[ix], axis[iy]])
        if sort_list:
            pts = pts[np.argsort(np.linalg.norm(pts - sites[i], axis=1))]
        lists.append(pts)
    return lists


_NATIVE_SWEEPS = (optwfr2, wfr2_grad_opt)


# --------------------------------------------------------------------------- f-3
def fftbounds(n, d=1):
    """Frequency bin edges of an fftshift-ed axis (imagetools.py:22-26)."""
    r = np.fft.fftshift(np.fft.fftfreq(n, d))
    return np.append(r, r[-1] + 1 / (n * d))


_THRESHOLD_FLOOR = 0.001


def _lower_threshold(t):
    """Next value of the reference's threshold schedule (geometric_phase_analysis.py:388-394): steps of 0.1
    down to 0.2, then halving, frozen once at or below 0.001.  Returns None when frozen."""
    if t <= _THRESHOLD_FLOOR:
        return None
    return t - 0.1 if t >= 0.2 else t / 2


def smallest_sum(ks):
    """Shortest of -a+b+c, a-b+c, a+b-c for three k-vectors a, b, c -- zero for a closed triangle
    (geometric_phase_analysis.py:538-548); NaN for anything but three vectors."""
    ks = np.asarray(ks, dtype=np.float64)
    if len(ks) != 3:
        return np.nan
    a, b, c = ks
    flipped = np.stack([-a + b + c, a - b + c, a + b - c])
    return flipped[np.argmin(np.sqrt((flipped * flipped).sum(axis=1)))]


def select_closest_to_triangle(ks):
    """The three k-vectors closest to forming a triangle (geometric_phase_analysis.py:529-535): the
    3-subset, first in itertools.combinations order, whose `smallest_sum` is shortest.  Closed
    triangles of grid frequencies tie at rounding level, so the lengths are taken exactly like the
    reference takes them (np.linalg.norm of the 2-vector)."""
    from itertools import combinations
    ks = np.asarray(ks, dtype=np.float64)
    best, best_defect = None, np.inf
    for triple in combinations(range(len(ks)), 3):
        defect = np.linalg.norm(smallest_sum(ks[list(triple)]))
        if defect < best_defect:
            best, best_defect = triple, defect
    return ks[list(best)]


def _peak_candidates(find, shape, sigma, dog, threshold, pix_norm_range):
    """One evaluation of the array work of extract_primary_ks (device: gpa_find_peaks; `find(sigma, dog_sigma,
    threshold)` -> (peaks, heights)) plus the ring selection and the +k / -k merge: (all_ks, pixel radii of the
    selected peaks, their heights)."""
    peaks, heights = find(sigma, 50.0 if dog else 0.0, threshold)
    offset = peaks - np.array(shape) // 2
    radius = np.sqrt((offset * offset).sum(axis=1))
    ring = (radius > pix_norm_range[0]) & (radius < pix_norm_range[1])
    peaks, radius, heights = peaks[ring], radius[ring], heights[ring]
    freq = [fftbounds(n) for n in shape]
    ks = np.stack([freq[0][peaks[:, 0]], freq[1][peaks[:, 1]]], axis=1) if len(peaks) else np.zeros((0, 2))
    return remove_negative_duplicates(ks), radius, heights


def _relaxed_parameters(n_found, radius, heights, threshold, sigma):
    """The reference's answer to "fewer than three k-vectors" (geometric_phase_analysis.py:447-467) as a
    pure function: the (threshold, sigma) to try next, or None when nothing is left to relax.
    With no peak at all only the threshold can drop; otherwise, in this order: a peak closer to the
    centre than 5 sigma narrows the smoothing to a sixth of that distance, a threshold above a fifth
    of the highest selected peak drops to it, else the threshold follows its schedule."""
    if n_found > 0:
        nearest = radius.min()
        if nearest < 5 * sigma:
            return threshold, nearest / 6
        fifth = 0.2 * np.max(heights)
        if threshold > fifth:
            return fifth, sigma
    lower = _lower_threshold(threshold)
    return None if lower is None else (lower, sigma)


def extract_primary_ks(image, plot=False, threshold=0.7, pix_norm_range=(2, 200), sigma=1, NMPERPIXEL=1., DoG=True,
                       dtype=None):
    """Primary k-vectors of an image from the smoothed Fourier transform of its periodic component
    (geometric_phase_analysis.py:397-505); returns (primary_ks, all_ks).

    The array work -- smooth + periodic DFT, |fftshift|, Gaussian / difference-of-Gaussians smoothing,
    peak_local_max -- runs on the device (gpa_find_peaks).  The reference adapts (threshold, sigma) by
    calling itself; every such call hands its result through unchanged, so the recursion is a chain,
    written here as a relaxation loop: evaluate, and while fewer than three k-vectors come out move
    to the next parameters (`_relaxed_parameters`; like the reference, every retry uses DoG=True).
    Then: three k-vectors are the answer, more than three are reduced to the triple closest to a
    triangle, fewer are returned as found.  ``plot`` only enables progress messages (no figure)."""
    image = np.asarray(image)
    plan = _plan(image, 1, dtype)
    return _primary_ks(_reusing(plan, lambda s, d, t: plan.find_peaks(image, s, d, t)), image.shape, plot, threshold,
                       pix_norm_range, sigma, DoG)


def extract_primary_ks_dev(plan, image_ptr, plot=False, threshold=0.7, pix_norm_range=(2, 200), sigma=1, DoG=True):
    """extract_primary_ks of an image that is resident on the device (`plan`: a _lib.Plan of its shape, `image_ptr`: device
    pointer to n0 x n1 reals of the plan's dtype; not modified): only the candidate lists -- a handful of (row, column,
    height) entries per evaluation -- come to the host."""
    return _primary_ks(_reusing(plan, lambda s, d, t: plan.find_peaks_dev(image_ptr, s, d, t)), plan.shape, plot, threshold,
                       pix_norm_range, sigma, DoG)


def _reusing(plan, find):
    """`find` for one image: an evaluation that differs from the previous one in the threshold only takes the smoothed spectrum
    the plan still holds (gpa_find_peaks_again: one threshold + maxima pass instead of transforms and filters)"""
    last = []

    def evaluate(sigma, dog_sigma, threshold):
        if last == [sigma, dog_sigma] and hasattr(plan, 'find_peaks_again'):
            return plan.find_peaks_again(threshold)
        out = find(sigma, dog_sigma, threshold)
        last[:] = [sigma, dog_sigma]
        return out
    return evaluate


def _primary_ks(find, shape, plot, threshold, pix_norm_range, sigma, DoG):
    dog = bool(DoG)
    while True:
        all_ks, radius, heights = _peak_candidates(find, shape, sigma, dog, threshold, pix_norm_range)
        if len(all_ks) >= 3:
            break
        step = _relaxed_parameters(len(all_ks), radius, heights, threshold, sigma)
        if step is None:
            if plot:
                print('extract_primary_ks: only %d k-vector(s) at threshold %.4g, sigma %.3g; giving up'
                      % (len(all_ks), threshold, sigma))
            break
        (threshold, sigma), dog = step, True
    primary_ks = select_closest_to_triangle(all_ks) if len(all_ks) > 3 else all_ks.copy()
    if plot and len(all_ks) > 3:
        print('extract_primary_ks: %d candidates, kept the triple closest to a triangle' % len(all_ks))
    return primary_ks, all_ks


# ------------------------------------------------------------- k-vector helpers (host, P x 2)
def average_lattice_vector(ks, symmetry=6):
    """Mean-radius vector at the periodic-mean angle (geometric_phase_analysis.py:303-306)."""
    ks = np.asarray(ks, dtype=np.float64)
    dt = periodic_average(np.arctan2(ks[:, 1], ks[:, 0]), period=2 * np.pi / symmetry)
    return np.linalg.norm(ks, axis=1).mean() * np.array([np.cos(dt), np.sin(dt)])


def calc_diff_from_isotropic(ani_ks, symmetry=6):
    """dks such that ani_ks + dks is an isotropic lattice (geometric_phase_analysis.py:309-322)."""
    ani_ks = np.asarray(ani_ks, dtype=np.float64)
    k_hex = average_lattice_vector(ani_ks, symmetry=symmetry)
    th = np.arctan2(k_hex[1], k_hex[0]) + 2 * np.pi / symmetry * np.arange(symmetry)
    ks_hex = np.linalg.norm(k_hex) * np.stack([np.cos(th), np.sin(th)], axis=-1)
    alldiffs = ks_hex - ani_ks[:, None]
    argmins = np.linalg.norm(alldiffs, axis=-1).argmin(axis=1)
    return alldiffs[np.arange(len(ani_ks)), argmins]


# --------------------------------------------------------------------------- a9
def per(image, inverse_dft=True, dtype=None):
    """Moisan's periodic + smooth decomposition, the third-party `moisan2011.per` the reference imports
    (geometric_phase_analysis.py:9; its one call is `pd, _ = per(image, inverse_dft=False)` at :429).
    inverse_dft=True (moisan2011's default): (p, s), the periodic and the smooth component, p + s = image;
    inverse_dft=False: their DFTs (p_hat, s_hat).  Computed on the device (`gpa_per`).  Parity of this function is
    unpinned (moisan2011 is not part of the reference checkout): restated from Moisan (2011) and held to the
    decomposition's defining properties (tests/per_properties.py)."""
    return _plan(image, 1, dtype).per(image, inverse_dft=bool(inverse_dft))


# --------------------------------------------------------------------------- a8
def fit_delta_k(phases):
    """Plane-fit slope of an unwrapped phase map as a k-vector correction
    (geometric_phase_analysis.py:92-94)."""
    return fit_plane(phases)[:2] / (2 * np.pi)


def iterate_GPA(image, kvecs, sigma, edge=5, iters=3, kmax_iter=25, kmax=200, verbose=False, dtype=None):
    """Iterate GPA, refining the reference vectors towards the extracted average
    (geometric_phase_analysis.py:116-154).  Lock-ins and weighted unwraps run on the
    device, including the 3-parameter Huber plane fit (gpa_fit_plane).

    Returns (prs, w, corr) like the reference."""
    from . import phase_unwrap as _pu
    image = np.asarray(image)
    kvecs = np.asarray(kvecs, dtype=np.float64).reshape(-1, 2)
    corr = np.zeros_like(kvecs)
    plan = _plan(image, len(kvecs), dtype)
    for i in range(iters + 1):
        rs = plan.lockin_batch(image, kvecs + corr, sigma)
        sl = (slice(edge, -edge), slice(edge, -edge)) if edge > 0 else (slice(None), slice(None))
        prs = [np.angle(r)[sl] for r in rs]
        w = np.stack([np.abs(r)[sl] for r in rs])
        last = i == iters
        prs = [_pu.phase_unwrap(r, np.sqrt(we / we.max()), kmax=kmax if last else kmax_iter, dtype=dtype)
               for r, we in zip(prs, w)]
        if not last:
            delta_ks = np.stack([fit_delta_k(pr) for pr in prs])
            if verbose:
                print(delta_ks)
            corr -= delta_ks
    return np.stack(prs), w, corr


def myweighed_lstsq(b, K, w, dtype=None):
    """Per-pixel weighted least squares, minimise ||w (K x - b)|| over x (2,) at every pixel
    (geometric_phase_analysis.py:97-113; imported by the reference's own property_extract.py:10): b (P, N, M'), K (P, 2)
    = 2 pi kvecs, w (P, N', M'') with N' >= N, M'' >= M' -- the reference indexes w[:, i, j] for the pixels of b, so the
    weight of a difference is that of its first pixel (:110) -- returns (2, N, M').  Device: gpa_weighted_lstsq
    (2 x 2 normal equations per pixel; the minimum-norm answer where they are singular, like np.linalg.lstsq)."""
    b = np.asarray(b)
    K = np.asarray(K, dtype=np.float64).reshape(-1, 2)
    w = np.asarray(w)
    if b.ndim != 3 or len(K) != b.shape[0] or w.ndim != 3 or w.shape[0] != b.shape[0]:
        raise ValueError('b (P, N, M), K (P, 2), w (P, >= N, >= M) expected')
    if w.shape[1] < b.shape[1] or w.shape[2] < b.shape[2]:
        raise ValueError('w must cover the pixels of b')
    plan = _lib.get_plan(b.shape[1:], len(K), DEFAULT_DTYPE if dtype is None else dtype)
    return plan.weighted_lstsq(b, w[:, :b.shape[1], :b.shape[2]], K / (2 * np.pi))


def reconstruct_u_inv(kvecs, b, weights=None, use_only_ks=None, dtype=None):
    """Unwrapped phases -> displacement field (geometric_phase_analysis.py:157-193).
    The weighted per-pixel solve runs on the device; the two global solves are one small
    matrix product on the host, as in the reference."""
    kvecs = np.asarray(kvecs, dtype=np.float64).reshape(-1, 2)
    K = 2 * np.pi * kvecs
    b = np.asarray(b, dtype=np.float64)
    b = b - b.mean(axis=(1, 2), keepdims=True)
    if use_only_ks is not None:
        assert len(use_only_ks) == 2
        idx = list(use_only_ks)
        us = np.linalg.inv(K[idx]) @ b[idx].reshape((2, -1))
        return us.reshape((2,) + b[0].shape)
    if weights is None:
        us = np.linalg.pinv(K) @ b.reshape((len(kvecs), -1))
        return us.reshape((2,) + b[0].shape)
    plan = _lib.get_plan(b.shape[1:], len(kvecs), DEFAULT_DTYPE if dtype is None else dtype)
    return plan.weighted_lstsq(b, weights, kvecs)


# --------------------------------------------------------------------------- a5 .. a7
def reconstruct_u_inv_from_phases(kvecs, phases, weights, weighted_unwrap=True, pre_diff=False, dtype=None):
    """Wrapped phases -> displacement field (geometric_phase_analysis.py:196-245).

    The device kernel consumes complex lock-ins, so phases/weights are re-packed as
    weights * exp(i phases); with mask border 0 the per-pixel least squares and the
    unwrap weights are those of the reference up to a uniform factor (1 + 1e-6)
    that neither solution depends on."""
    phases = np.asarray(phases, dtype=np.float64)
    weights = np.asarray(weights, dtype=np.float64)
    plan = _lib.get_plan(weights.shape[1:], len(weights), DEFAULT_DTYPE if dtype is None else dtype)
    if pre_diff:
        # phases is (P, N, M, 2): given phase gradients, wrapped and solved per pixel on the device (:228-232)
        dudx, dudy, wnorm = plan.reconstruct_prediff(phases, weights, kvecs)
    else:
        dudx, dudy, wnorm = plan.reconstruct_grad(weights * np.exp(1j * phases), kvecs, 0)
    us = []
    for i in range(2):
        if weighted_unwrap:
            us.append(plan.unwrap_prediff(dudx[i], dudy[i], wnorm, kmax=10)[0])
        else:
            us.append(plan.unwrap_prediff(dudx[i], dudy[i])[0])
    return np.array(us)


def extract_displacement_field(image, kvecs, sigma=None, kwscale=2.5, ksteps=3, return_gs=False,
                               wfr_func=optwfr2, deconvolve=False, klists=None, dtype=None):
    """Top level convenience function (geometric_phase_analysis.py:907-932).

    With the package's own sweep functions as ``wfr_func`` the whole chain (mean
    subtraction, P x K lock-ins, selection, phases/weights, per-pixel least squares,
    two weighted unwraps) runs in one fused device call; any other callable is
    invoked per peak exactly like the reference does and only the reconstruction
    runs on the device.  ``klists`` (P lists of (K,2)) replaces the np.arange grid.
    ``deconvolve=True`` Wiener-deconvolves u with the lock-in Gaussian afterwards (:927-928).
    """
    image = np.asarray(image)
    kvecs = np.asarray(kvecs, dtype=np.float64).reshape(-1, 2)
    norms = np.linalg.norm(kvecs, axis=1)
    kw = norms.mean() / kwscale
    if sigma is None:
        sigma = int(np.ceil(1 / norms.min()))
    kstep = kw / ksteps
    dr = int(2 * sigma)
    if wfr_func in _NATIVE_SWEEPS and not (return_gs and wfr_func is wfr2_grad_opt):
        if klists is None:
            klists = [_sweep_list(pk[0], pk[1], kw, kstep) for pk in kvecs]
        klists = [np.asarray(kl, dtype=np.float64).reshape(-1, 2) for kl in klists]
        K = max(len(kl) for kl in klists)
        # lists of unequal length: repeat the last candidate (a repeat can never win a strict '>')
        padded = np.stack([np.concatenate([kl, np.repeat(kl[-1:], K - len(kl), axis=0)]) for kl in klists])
        plan = _lib.get_plan(image.shape, len(kvecs) * K, DEFAULT_DTYPE if dtype is None else dtype)
        u, lock, kidx, _ = plan.extract_displacement_field(image, kvecs, padded, sigma, dr, kmax=10,
                                                           want_lockins=return_gs, want_kidx=return_gs)
        if deconvolve:
            u = gaussian_deconvolve(u, sigma, dr, dtype=dtype)
        if return_gs:
            gs = [{'lockin': lock[p], 'w': _w_from_kidx(kidx[p], padded[p]), 'kidx': kidx[p]}
                  for p in range(len(kvecs))]
            return u, gs
        return u
    gs = [wfr_func(image - image.mean(), sigma, pk[0], pk[1], kw=kw, kstep=kstep) for pk in kvecs]
    lockins = np.stack([g['lockin'] for g in gs])
    plan = _lib.get_plan(image.shape, len(kvecs), DEFAULT_DTYPE if dtype is None else dtype)
    dudx, dudy, wnorm = plan.reconstruct_grad(lockins, kvecs, dr)
    u = np.array([plan.unwrap_prediff(dudx[i], dudy[i], wnorm, kmax=10)[0] for i in range(2)])
    if deconvolve:
        u = gaussian_deconvolve(u, sigma, dr, dtype=dtype)
    if return_gs:
        return u, gs
    return u


def extract_displacement_field_stack(images, kvecs, sigma=None, kwscale=2.5, ksteps=3, klists=None, dtype=None):
    """`extract_displacement_field` of every image of a stack (B, N, M) with one set of k-vectors -- frames of a
    movie, a tilt or temperature series -- in ONE device call: every kernel of the driver takes an image index from
    its grid, so the ~110 dependent launches that bound a small image are paid once per stack (512^2 frames: x4
    the single-image rate).  No counterpart in the reference (a Python loop over `extract_displacement_field`,
    geometric_phase_analysis.py:907-932, gives the same numbers).
    Returns u of shape (B, 2, N, M)."""
    images = np.asarray(images)
    if images.ndim != 3:
        raise ValueError('images must be a stack (B, N, M)')
    kvecs = np.asarray(kvecs, dtype=np.float64).reshape(-1, 2)
    norms = np.linalg.norm(kvecs, axis=1)
    kw = norms.mean() / kwscale
    if sigma is None:
        sigma = int(np.ceil(1 / norms.min()))
    if klists is None:
        klists = [_sweep_list(pk[0], pk[1], kw, kw / ksteps) for pk in kvecs]
    klists = [np.asarray(kl, dtype=np.float64).reshape(-1, 2) for kl in klists]
    K = max(len(kl) for kl in klists)
    padded = np.stack([np.concatenate([kl, np.repeat(kl[-1:], K - len(kl), axis=0)]) for kl in klists])
    plan = _lib.get_plan(images.shape[1:], len(kvecs) * K, DEFAULT_DTYPE if dtype is None else dtype)
    return plan.extract_displacement_field_stack(images, kvecs, padded, sigma, int(2 * sigma), kmax=10)[0]


def gaussian_deconvolve(data, sigma, dr=20, balance=5000, dtype=None):
    """Deconvolve a stack of fields by the Gaussian of width sigma (geometric_phase_analysis.py:892-904):
    reflect padding by 2*dr, skimage.restoration.wiener with `balance`, cropping -- on the device
    (gpa_gaussian_deconvolve), one field at a time like the reference."""
    data = np.asarray(data)
    if data.ndim < 2:
        raise ValueError('data must have at least 2 dimensions')
    dr = int(dr)
    m0, m1 = data.shape[-2:]
    plan = _lib.get_plan((m0 + 4 * dr, m1 + 4 * dr), 1, DEFAULT_DTYPE if dtype is None else dtype)
    dec = [plan.gaussian_deconvolve(f, dr, sigma, balance) for f in data.reshape((-1, m0, m1))]
    return np.reshape(np.stack(dec), data.shape)


# --------------------------------------------------------------------------- f-1
def invert_u_overlap(us, iters=35, edge=0, mode='nearest', dtype=None):
    """Numerical inverse of the displacement `us` (geometric_phase_analysis.py:262-300):
    u_it(r + us(r)) = r, by `iters` rounds of cubic-spline resampling on the grid extended by
    `edge` pixels.  mode: scipy.ndimage's 'nearest' (the reference's default) or 'constant' (cval 0, the last round
    cval = nan as in the reference); the other scipy modes are not provided."""
    us = np.asarray(us)
    plan = _lib.get_plan(us.shape[1:], 1, DEFAULT_DTYPE if dtype is None else dtype)
    return plan.invert_u_overlap(us, iters=iters, edge=edge, mode=mode)


def invert_u(us, iters=35, edge=0, mode='nearest', dtype=None):
    """The variant without overlap (geometric_phase_analysis.py:248-259): u_it on the image's own grid; mode 'nearest'
    or 'constant' as above."""
    us = np.asarray(us)
    plan = _lib.get_plan(us.shape[1:], 1, DEFAULT_DTYPE if dtype is None else dtype)
    return plan.invert_u(us, iters=iters, edge=edge, mode=mode)


def undistort_image(deformed, u, dtype=None):
    """Reconstruct an undistorted image from `deformed` and the displacement field `u`
    (Lawler-Fujita, geometric_phase_analysis.py:935-974)."""
    deformed = np.asarray(deformed)
    plan = _lib.get_plan(deformed.shape, 1, DEFAULT_DTYPE if dtype is None else dtype)
    return plan.undistort_image(deformed, u)
