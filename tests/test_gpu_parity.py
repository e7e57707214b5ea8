"""GPU parity tests: the HIP path (through the C ABI) against the golden vectors made
by the real reference and against the CPU oracle.  Run with `-m gpu` on an MI355X.

Stated tolerances (relative to the largest magnitude of the compared array):
    f64 path: 1e-11 for lock-ins / gradients, 1e-9 for PCG outputs
    f32 path: 2e-6 for lock-ins, 2e-5 for gradients, 2e-5 for PCG outputs
Index / mask outputs (kidx, 'w') are compared bit-exactly; a mismatch is tolerated
only where the two candidates' amplitudes tie to rounding (kidx_mismatch_is_tie)."""
import os

import numpy as np
import pytest

from conftest import kidx_mismatch_is_tie
from oracle import gpa_oracle as orc
from pygpa_amd import _lib
from pygpa_amd.synthetic import hex_kvecs, gaussian_bump_displacement, hex_moire, explicit_klists

pytestmark = pytest.mark.gpu

TOL = {
    np.float64: dict(lock=1e-11, grad=1e-11, pcg=1e-9, tie=1e-12),
    np.float32: dict(lock=2e-6, grad=2e-5, pcg=2e-5, tie=2e-6),
}
DTYPES = [np.float64, np.float32]
FULL_CASES = ['hex_64', 'hex_48x80', 'hex_63x65', 'hex_60']


def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def check_kidx(kidx, ref_kidx, img0, klist, sigma, tie_tol):
    bad = kidx != ref_kidx
    if not bad.any():
        return
    amps = np.abs(orc.lockin_batch(img0, klist, sigma))
    assert np.all(kidx_mismatch_is_tie(amps, kidx, ref_kidx, tie_tol)[bad]), \
        '%d kidx mismatches that are not amplitude ties' % int(bad.sum())


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('name', FULL_CASES)
def test_a1_a2_lockin(golden, name, dtype):
    g = golden(name)
    img0 = g['image'] - g['image'].mean()
    plan = _lib.Plan(img0.shape, 3, dtype)
    out = plan.lockin_batch(img0, g['kvecs'][:2], int(g['sigma']))
    assert out.dtype == (np.complex128 if dtype is np.float64 else np.complex64)
    assert rel(out[0], g['a1_GPA']) < TOL[dtype]['lock']
    assert rel(out[1], g['a1_optGPA']) < TOL[dtype]['lock']
    if 'a2_vecGPA' in g:
        out3 = plan.lockin_batch(img0, g['kvecs'], int(g['sigma']))
        assert rel(out3, g['a2_vecGPA']) < TOL[dtype]['lock']
    plan.close()


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('name', FULL_CASES)
def test_a3_sweep(golden, name, dtype):
    g = golden(name)
    img0 = g['image'] - g['image'].mean()
    sigma = int(g['sigma'])
    K = g['a3_klists'].shape[1]
    plan = _lib.Plan(img0.shape, K, dtype)
    for p in range(3):
        lock, kidx, _ = plan.sweep(img0, g['kvecs'][p], g['a3_klists'][p], sigma)
        check_kidx(kidx, g['a3_kidx'][p], img0, g['a3_klists'][p], sigma, TOL[dtype]['tie'])
        same = kidx == g['a3_kidx'][p]
        assert rel(lock[same], g['a3_lockin'][p][same]) < TOL[dtype]['lock']
        if dtype is np.float64:
            assert np.array_equal(kidx, g['a3_kidx'][p])          # index work: bit-exact in f64
    plan.close()


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('name', FULL_CASES)
def test_a5_a6_reconstruct(golden, name, dtype):
    g = golden(name)
    plan = _lib.Plan(g['image'].shape, 3, dtype)
    dudx, dudy, wn = plan.reconstruct_grad(g['a3_lockin'], g['kvecs'], 2 * int(g['sigma']))
    assert rel(dudx, g['a6_dudx']) < TOL[dtype]['grad']
    assert rel(dudy, g['a6_dudy']) < TOL[dtype]['grad']
    assert rel(wn, np.linalg.norm(g['a5_weights'], axis=0)) < TOL[dtype]['lock']
    plan.close()


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('name', FULL_CASES)
def test_a7_unwrap_golden(golden, name, dtype):
    """power-of-two sizes run the Makhoul/FFT DCT kernels, the 48x80 and 63x65 cases the
    Bluestein ones (and pin the reference's swapped-axis eigenvalues on non-square images)"""
    g = golden(name)
    plan = _lib.Plan(g['image'].shape, 1, dtype)
    wn = np.linalg.norm(g['a5_weights'], axis=0)
    for kmax in (1, 3, 10, 100):
        phi, it = plan.unwrap_prediff(g['a6_dudx'][0], g['a6_dudy'][0], wn, kmax=kmax)
        # (kmax = 100 in f32: the solve runs to the f32 rounding floor, where the stagnation guard ends it -- README
        #  "Tolerances"; measured 2.1e-5 of max |phi| on hex_60, below 1e-5 on the other cases)
        tol = TOL[dtype]['pcg'] * (2.5 if (dtype is np.float32 and kmax == 100) else 1.0)
        assert rel(phi, g['a7_phi_w_kmax%d' % kmax]) < tol, kmax
        if dtype is np.float64 and name == 'hex_64':
            assert it == min(kmax, 15)      # the reference converges in 15 iterations on this case
    phi, it = plan.unwrap_prediff(g['a6_dudx'][0], g['a6_dudy'][0])
    assert rel(phi, g['a7_phi_unweighted']) < TOL[dtype]['pcg']
    w0 = g['a5_weights'][0]
    psi, _ = plan.unwrap(g['a5_phases'][0], np.sqrt(w0 / w0.max()), kmax=10)
    assert rel(psi, g['a7_psi_unwrap_kmax10']) < TOL[dtype]['pcg']
    plan.close()


@pytest.mark.parametrize('dtype', DTYPES)
def test_reference_unwrap_ramp(golden, dtype):
    """The reference's own unwrap test (tests/test_phase_unwrap.py) at 64^2 against its outputs."""
    g = golden('unwrap_ramp_64')
    plan = _lib.Plan(g['psi'].shape, 1, dtype)
    atol = 1e-8 if dtype is np.float64 else 2e-4
    for kmax in (1, 5, 30):
        phi, _ = plan.unwrap(g['psi'], np.ones_like(g['psi']), kmax=kmax)
        assert np.allclose(phi, g['ref_kmax%d' % kmax], atol=atol)
        assert np.allclose(phi - phi.mean(), g['psi0'] - g['psi0'].mean(), atol=atol)
        phi_u, _ = plan.unwrap(g['psi'], None, kmax=kmax)
        assert np.allclose(phi_u, phi, atol=atol)
    phi, _ = plan.unwrap(g['psi'], g['gaussian_weight'])
    assert np.allclose(phi, g['ref_gaussian'], atol=10 * atol)
    plan.close()


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('name', ['hex_64', 'hex_48x80', 'hex_63x65', 'hex_128_noise', 'hex_60'])
def test_fused_driver_golden(golden, name, dtype):
    g = golden(name)
    sigma = int(g['sigma'])
    K = g['a3_klists'].shape[1]
    plan = _lib.Plan(g['image'].shape, 3 * K, dtype)
    u, lock, kidx, iters = plan.extract_displacement_field(g['image'], g['kvecs'], g['a3_klists'], sigma, 2 * sigma,
                                                            kmax=10, want_lockins=True, want_kidx=True)
    img0 = g['image'] - g['image'].mean()
    for p in range(3):
        check_kidx(kidx[p], g['a3_kidx'][p], img0, g['a3_klists'][p], sigma, TOL[dtype]['tie'])
    assert rel(u, g['u']) < TOL[dtype]['pcg']
    if dtype is np.float64:
        assert np.array_equal(kidx, g['a3_kidx'])
        assert iters == (10, 10)
    plan.close()


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('shape', [(256, 512), (200, 300), (1024, 1024)])
def test_sweep_vs_oracle_larger(shape, dtype):
    kvecs = hex_kvecs(0.1, 7.0)
    img = hex_moire(shape, kvecs, gaussian_bump_displacement(shape), noise=0.2, seed=11)
    img0 = img - img.mean()
    kw, sigma, _ = orc.derive_params(kvecs)
    klist = explicit_klists(kvecs, kw, 3, 3)[1]
    ref = orc.sweep(img0, sigma, klist, kvecs[1], workers=8)
    plan = _lib.Plan(shape, len(klist), dtype)
    lock, kidx, _ = plan.sweep(img0, kvecs[1], klist, sigma)
    check_kidx(kidx, ref['kidx'], img0, klist, sigma, TOL[dtype]['tie'])
    same = kidx == ref['kidx']
    assert same.mean() > 0.999
    assert rel(lock[same], ref['lockin'][same]) < TOL[dtype]['lock']
    plan.close()


@pytest.mark.parametrize('dtype', DTYPES)
def test_driver_vs_oracle_512(dtype):
    """Config-1-sized image: recovered displacement within the reference test's 0.9 px bar
    (tests/test_geometric_phase_analysis.py:61-66) and equal to the oracle."""
    shape = (512, 512)
    kvecs = hex_kvecs(0.1, 7.0)
    u_true = gaussian_bump_displacement(shape)
    img = hex_moire(shape, kvecs, u_true, noise=0.3, seed=3)
    u_ref, parts = orc.extract_displacement_field(img, kvecs, return_parts=True, workers=8)
    import pygpa_amd.geometric_phase_analysis as GPA
    u = GPA.extract_displacement_field(img, kvecs, dtype=dtype)
    assert rel(u, u_ref) < (1e-8 if dtype is np.float64 else 5e-4)
    assert np.all(np.abs(-u - u_true)[:, 20:-20, 20:-20] < 0.9)


def test_full_size_properties_4096():
    """BASELINE size (4096^2, f32): properties that need no oracle run at full size."""
    n = 4096
    kvecs = hex_kvecs(0.1, 7.0)
    img = hex_moire((n, n), kvecs, gaussian_bump_displacement((n, n)), noise=0.1, seed=2, dtype=np.float32)
    kw, sigma, _ = orc.derive_params(kvecs)
    klists = np.stack(explicit_klists(kvecs, kw, 2, 2))
    plan = _lib.Plan((n, n), 12, np.float32)
    # (1) linearity of the lock-in
    a = plan.lockin_batch(img, kvecs[:1], sigma)[0]
    b = plan.lockin_batch(2.5 * img, kvecs[:1], sigma)[0]
    assert rel(b, 2.5 * a) < 1e-5
    # (2) a sweep over a single candidate equal to kref is the plain lock-in
    lock, kidx, _ = plan.sweep(img, kvecs[0], kvecs[:1], sigma)
    assert np.array_equal(kidx, np.zeros_like(kidx))
    assert rel(lock, a) < 1e-6
    # (3) one row/column block against the oracle (full 2-D oracle at 4096^2 takes seconds)
    ref = orc.lockin(img.astype(np.float64), kvecs[0], sigma, workers=8)
    assert rel(a, ref) < 5e-6
    # (4) the driver is invariant to an image offset (mean subtraction, reference :919) and
    #     reproducible run to run (deterministic reductions)
    u1, _, k1, it1 = plan.extract_displacement_field(img, kvecs, klists, sigma, 2 * sigma, want_kidx=True)
    u2, _, k2, it2 = plan.extract_displacement_field(img, kvecs, klists, sigma, 2 * sigma, want_kidx=True)
    assert np.array_equal(u1, u2) and np.array_equal(k1, k2)
    u3, _, k3, _ = plan.extract_displacement_field(img + np.float32(0.5), kvecs, klists, sigma, 2 * sigma, want_kidx=True)
    assert (k3 != k1).mean() < 1e-4
    assert rel(u3, u1) < 1e-2
    assert np.isfinite(u1).all()
    plan.close()


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('name', FULL_CASES)
def test_a4_grad(golden, name, dtype):
    g = golden(name)
    img0 = g['image'] - g['image'].mean()
    sigma = int(g['sigma'])
    K = g['a3_klists'].shape[1]
    plan = _lib.Plan(img0.shape, K, dtype)
    lock, kidx, grad = plan.sweep(img0, g['kvecs'][0], g['a3_klists'][0], sigma, want_grad=True)
    check_kidx(kidx, g['a3_kidx'][0], img0, g['a3_klists'][0], sigma, TOL[dtype]['tie'])
    same = kidx == g['a3_kidx'][0]
    assert rel(lock[same], g['a3_lockin'][0][same]) < TOL[dtype]['lock']
    # compare modulo the pi-periodic wrap of wrapToPi(2 g) / 2; the phase of a weak
    # lock-in is ill-conditioned, so weigh the f32 comparison by the local amplitude
    d = orc.wrap_to_pi(2 * (grad.astype(np.float64) - g['a4_grad0'])) / 2
    amp = np.abs(g['a3_lockin'][0])
    ok = same & (amp > 1e-3 * amp.max())
    assert np.abs(d[ok]).max() < (1e-9 if dtype is np.float64 else 2e-3)
    plan.close()


def test_a8_iterate_gpa(golden):
    """iterate_GPA / reconstruct_u_inv against the reference's outputs (64^2 image, unwraps on
    the 54^2 cropped maps -> Bluestein path, kmax_iter = 25, final kmax = 200)."""
    import pygpa_amd.geometric_phase_analysis as GPA
    g = golden('iterate_64')
    prs, w, corr = GPA.iterate_GPA(g['image'] - g['image'].mean(), g['start_ks'], int(g['sigma']))
    assert np.allclose(corr, g['corr'], rtol=1e-5, atol=1e-8)
    assert np.allclose(w, g['w'], rtol=1e-8, atol=1e-10)
    assert np.allclose(prs, g['prs'], rtol=1e-5, atol=1e-6)
    assert np.abs(g['start_ks'] + corr - g['true_ks']).max() < 5e-4
    uw = GPA.reconstruct_u_inv(g['start_ks'] + g['corr'], g['prs'], weights=g['w'])
    assert np.allclose(uw, g['u_weighted'], atol=1e-9)
    ug = GPA.reconstruct_u_inv(g['start_ks'] + g['corr'], g['prs'])
    assert np.allclose(ug, g['u_global'], atol=1e-9)
    # the single-precision build of the same route: lock-ins, unwraps (kmax 25 / 200 hit the f32 residual floor
    # first), Huber plane fits and the per-pixel solve in f32; the refined k-vectors land on the true ones as well
    prs32, w32, corr32 = GPA.iterate_GPA(g['image'] - g['image'].mean(), g['start_ks'], int(g['sigma']), dtype=np.float32)
    assert np.allclose(corr32, g['corr'], rtol=0, atol=2e-6)
    assert np.allclose(w32, g['w'], rtol=2e-5, atol=1e-5)
    assert np.abs(prs32 - g['prs']).max() < 2e-3 * np.abs(g['prs']).max()
    assert np.abs(g['start_ks'] + corr32 - g['true_ks']).max() < 5e-4
    uw32 = GPA.reconstruct_u_inv(g['start_ks'] + g['corr'], g['prs'], weights=g['w'], dtype=np.float32)
    assert np.abs(uw32 - g['u_weighted']).max() < 2e-5 * max(1.0, np.abs(g['u_weighted']).max())


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('shape', [(64, 64), (48, 80), (63, 65), (500, 500)])
def test_a9_per_dft(shape, dtype):
    """periodic-component DFT (arbitrary-size 2-D DFT via Bluestein) against the oracle's
    restatement of Moisan (2011) and its known answers."""
    import pygpa_amd.geometric_phase_analysis as GPA
    rng = np.random.default_rng(5)
    img = rng.normal(size=shape) + np.linspace(0, 3, shape[1])[None, :] + np.linspace(-1, 0, shape[0])[:, None] ** 2
    img = img - img.mean()          # as the call site does (geometric_phase_analysis.py:428)
    phat, shat = GPA.per(img, inverse_dft=False, dtype=dtype)
    ref, sref = orc.per(img, inverse_dft=False)
    # f32: a chirp-z DFT carries ~1e-5 of the largest bin as error (f64: 1e-11)
    tol = 1e-10 if dtype is np.float64 else 5e-5   # the oracle's own 2cos+2cos-4 cancels to ~1e-11 near DC
    assert rel(phat, ref) < tol
    # known answers: mean(p) = mean(image) (= 0 here); p + s = image
    assert abs(phat[0, 0]) < tol * np.abs(ref).max()
    assert rel(phat + sref, np.fft.fft2(img)) < tol
    # the other three outputs of moisan2011.per: s_hat, and (inverse_dft=True, its default) the components themselves
    assert np.abs(shat - sref).max() < tol * np.abs(ref).max()
    pc, sc = GPA.per(img, dtype=dtype)
    pref, sref_c = orc.per(img, inverse_dft=True)
    assert pc.dtype == np.dtype(dtype) and sc.dtype == np.dtype(dtype)
    sc_tol = (1e-10 if dtype is np.float64 else 2e-5) * np.abs(img).max()
    assert np.abs(pc - pref).max() < sc_tol and np.abs(sc - sref_c).max() < sc_tol
    assert np.abs(pc + sc - img).max() < (1e-12 if dtype is np.float64 else 1e-6) * np.abs(img).max()


def test_tiled_path_device_vs_oracle():
    """tile-sharded path (pygpa_amd/distributed.py), one rank: device stages against the oracle
    run on the SAME tiling (2 x 2 tiles of 128 x 192 -> 104 x 136 windows: padded lock-ins,
    global 128 x 192 unwrap on the Bluestein/pow2 mix)."""
    from pygpa_amd import distributed as D
    from test_distributed import _case, _oracle_compute
    img, kvecs, klists = _case()
    u_ref = D.extract_displacement_field_tiled(img, kvecs, (2, 2), klists=klists, halo=20, compute=_oracle_compute())
    u = D.extract_displacement_field_tiled(img, kvecs, (2, 2), klists=klists, halo=20)
    assert rel(u, u_ref) < 1e-8
    u32 = D.extract_displacement_field_tiled(img, kvecs, (2, 2), klists=klists, halo=20, dtype=np.float32)
    assert rel(u32, u_ref) < 5e-4


@pytest.mark.parametrize('dtype', DTYPES)
def test_f1_lawler_fujita_golden(golden, dtype):
    """invert_u_overlap / undistort_image against the reference's outputs (96 x 80 field):
    scipy.ndimage.map_coordinates semantics (spline prefilter, mode='nearest' with its
    12-sample edge padding, mode='constant' for the final resampling) restated on the device."""
    import pygpa_amd.geometric_phase_analysis as GPA
    g = golden('warp_96x80')
    tol = 1e-10 if dtype is np.float64 else 2e-4
    u_inv = GPA.invert_u_overlap(-g['u'], dtype=dtype)
    assert np.abs(u_inv - g['u_inv']).max() < tol * max(1.0, np.abs(g['u_inv']).max())
    u_e = GPA.invert_u_overlap(-g['u'], iters=5, edge=4, dtype=dtype)
    assert u_e.shape == g['u_inv_edge4_it5'].shape
    assert np.abs(u_e - g['u_inv_edge4_it5']).max() < tol * max(1.0, np.abs(g['u_inv_edge4_it5']).max())
    rec = GPA.undistort_image(g['deformed'], g['u'], dtype=dtype)
    # mode='constant' is discontinuous where r + u_inv leaves [0, n-1]: in f32 a sample a few
    # ulps from that border may land on the other side, so compare away from it
    xx, yy = np.mgrid[:rec.shape[0], :rec.shape[1]]
    cx, cy = xx + g['u_inv'][0], yy + g['u_inv'][1]
    margin = 0.0 if dtype is np.float64 else 1e-3
    ok = (cx >= margin) & (cx <= rec.shape[0] - 1 - margin) & (cy >= margin) & (cy <= rec.shape[1] - 1 - margin)
    ok |= (cx < -margin) | (cx > rec.shape[0] - 1 + margin) | (cy < -margin) | (cy > rec.shape[1] - 1 + margin)
    d = np.abs(rec - g['reconstructed'])
    assert d[ok].max() < (1e-9 if dtype is np.float64 else 5e-4) * np.abs(g['reconstructed']).max()
    assert ok.mean() > 0.98


def test_f1_reconstruction_like_reference():
    """reference tests/test_geometric_phase_analysis.py:73-78: undistorting with the true
    displacement recovers the undeformed lattice within 2 % of its maximum"""
    import pygpa_amd.geometric_phase_analysis as GPA
    shape = (512, 512)
    ks = hex_kvecs(0.1, 7.0)
    u = gaussian_bump_displacement(shape)
    original = hex_moire(shape, ks)
    deformed = hex_moire(shape, ks, u)
    u_inv = GPA.invert_u_overlap(-u)
    assert u_inv.shape == u.shape
    rec = GPA.undistort_image(deformed, u)
    assert np.all(np.abs(rec - original)[2:-2, 2:-2] / np.abs(original).max() < 0.02)
    assert rel(rec, orc.undistort_image(deformed, u)) < 1e-9


class DeviceArray:
    """a device buffer through the HIP runtime itself (ctypes), so the test needs no torch"""
    _hip = None

    def __init__(self, host):
        import ctypes as C
        if DeviceArray._hip is None:
            _lib.load()
            DeviceArray._hip = C.CDLL('libamdhip64.so')
        self.host = np.ascontiguousarray(host)
        p = C.c_void_p()
        assert self._hip.hipMalloc(C.byref(p), C.c_size_t(self.host.nbytes)) == 0
        self.ptr = p.value
        assert self._hip.hipMemcpy(C.c_void_p(self.ptr), self.host.ctypes.data_as(C.c_void_p), C.c_size_t(self.host.nbytes), 1) == 0

    def get(self):
        import ctypes as C
        out = np.empty_like(self.host)
        assert self._hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr), C.c_size_t(out.nbytes), 2) == 0
        return out

    def __del__(self):
        import ctypes as C
        try:
            self._hip.hipFree(C.c_void_p(self.ptr))
        except Exception:
            pass


@pytest.mark.parametrize('dtype', DTYPES)
def test_f1_early_exit_equals_every_round(dtype, gpa_option):
    """The fixed point u_it(r) <- u(r + u_it(r)) (geometric_phase_analysis.py:291-299, :255-258) leaves a wavefront's loop once
    every pixel is at a bitwise fixed point or alternates between two values, and picks the member of such a cycle by the
    parity of the rounds left: BIT-identical to running every round (LF_ALL_ROUNDS=1), for even and odd round counts, both
    boundary modes, overlap (with its final cval = NaN round) and plain variant, the tile and the row kernel.  The field is
    large and rough enough that cycles occur (asserted: an even and an odd count differ somewhere in f32)."""
    shape = (600, 520)
    rng = np.random.default_rng(11)
    u = (2.0 * gaussian_bump_displacement(shape) + 0.02 * rng.normal(size=(2,) + shape)).astype(dtype)
    plan = _lib.Plan(shape, 1, dtype)
    got = {}
    for lftile in (None, '1'):
        for overlap, edge, mode in ((True, 0, 'nearest'), (False, 3, 'nearest'), (True, 4, 'constant'), (False, 0, 'constant')):
            for iters in (35, 36):
                gpa_option('NO_LFTILE', lftile)
                gpa_option('LF_ALL_ROUNDS', None)
                fn = plan.invert_u_overlap if overlap else plan.invert_u
                a = fn(u, iters=iters, edge=edge, mode=mode)
                gpa_option('LF_ALL_ROUNDS', '1')
                b = fn(u, iters=iters, edge=edge, mode=mode)
                gpa_option('LF_ALL_ROUNDS', None)
                assert np.array_equal(a, b, equal_nan=True), (lftile, overlap, edge, mode, iters, int((a != b).sum()))
                got[(lftile, overlap, edge, mode, iters)] = a
    gpa_option('NO_LFTILE', None)
    if dtype is np.float32:
        k35, k36 = (None, False, 3, 'nearest', 35), (None, False, 3, 'nearest', 36)
        assert not np.array_equal(got[k35], got[k36])      # cycles exist in this field: the parity matters
    plan.close()


@pytest.mark.parametrize('dtype', DTYPES)
def test_f1_device_entry_points_equal_host_entry_points(dtype):
    """round 5: gpa_invert_u_mode_dev / gpa_undistort_image_dev (device pointers, enqueued on the plan's stream, scratch
    kept by the plan, no host round trip of u) against the host-pointer entry points they now back: BIT-identical fields,
    both boundary modes, overlap and plain variant, the scale argument (undistort_image inverts -u); and the window
    argument: four windows of the output grid, computed one after the other, assemble the whole-grid result."""
    shape = (200, 168)
    rng = np.random.default_rng(5)
    ks = hex_kvecs(0.1, 7.0)
    u = (0.6 * gaussian_bump_displacement(shape) + 0.05 * rng.normal(size=(2,) + shape)).astype(dtype)
    deformed = hex_moire(shape, ks, u.astype(np.float64)).astype(dtype)
    plan = _lib.Plan(shape, 1, dtype)
    d_u, d_img = DeviceArray(u), DeviceArray(deformed)
    for overlap, edge, mode in ((True, 0, 'nearest'), (True, 6, 'nearest'), (False, 3, 'nearest'), (True, 4, 'constant'), (False, 0, 'constant')):
        host = plan.invert_u_overlap(u, iters=7, edge=edge, mode=mode) if overlap else plan.invert_u(u, iters=7, edge=edge, mode=mode)
        d_out = DeviceArray(np.zeros_like(host))
        plan.invert_u_dev(d_u.ptr, d_out.ptr, scale=1.0, iters=7, edge=edge, overlap=overlap, mode=mode)
        plan.sync()
        assert np.array_equal(d_out.get(), host, equal_nan=True), (overlap, edge, mode)
        # scale = -1 on the device = the host call on -u
        host_neg = plan.invert_u_overlap(-u, iters=7, edge=edge, mode=mode) if overlap else plan.invert_u(-u, iters=7, edge=edge, mode=mode)
        plan.invert_u_dev(d_u.ptr, d_out.ptr, scale=-1.0, iters=7, edge=edge, overlap=overlap, mode=mode)
        plan.sync()
        assert np.array_equal(d_out.get(), host_neg, equal_nan=True), (overlap, edge, mode)
    # undistort_image: whole grid, then as four windows (ragged split) into a poisoned buffer
    rec_host = plan.undistort_image(deformed, u)
    uinv_host = plan.invert_u_overlap(-u)
    d_rec, d_uinv = DeviceArray(np.full(shape, -7.0, dtype)), DeviceArray(np.full((2,) + shape, -7.0, dtype))
    plan.undistort_image_dev(d_img.ptr, d_u.ptr, d_rec.ptr, uinv_ptr=d_uinv.ptr)
    plan.sync()
    assert np.array_equal(d_rec.get(), rec_host) and np.array_equal(d_uinv.get(), uinv_host)
    d_rec2, d_uinv2 = DeviceArray(np.full(shape, -7.0, dtype)), DeviceArray(np.full((2,) + shape, -7.0, dtype))
    r_split, c_split = 77, 100
    plan.undistort_image_dev(d_img.ptr, d_u.ptr, d_rec2.ptr, uinv_ptr=d_uinv2.ptr, rects=(0, 0, r_split, c_split))
    plan.sync()
    part = d_rec2.get()
    assert np.array_equal(part[:r_split, :c_split], rec_host[:r_split, :c_split])
    assert np.all(part[r_split:] == -7.0) and np.all(part[:, c_split:] == -7.0)      # nothing outside the window is touched
    plan.undistort_image_dev(d_img.ptr, d_u.ptr, d_rec2.ptr, uinv_ptr=d_uinv2.ptr,          # three windows behind one prefilter
                             rects=[(0, c_split, r_split, shape[1] - c_split), (r_split, 0, shape[0] - r_split, c_split),
                                    (r_split, c_split, shape[0] - r_split, shape[1] - c_split)])
    plan.sync()
    assert np.array_equal(d_rec2.get(), rec_host) and np.array_equal(d_uinv2.get(), uinv_host)
    # a window that is empty or lies outside the grid is refused, not silently skipped; one that sticks out is clipped
    before = plan.workspace_bytes
    for bad in ((0, 0, 0, 5), (0, 0, 5, -1), (shape[0], 0, 4, 4), (0, shape[1] + 3, 4, 4), (-9, 0, 9, 4)):
        with pytest.raises(_lib.GPAError):
            plan.undistort_image_dev(d_img.ptr, d_u.ptr, d_rec2.ptr, uinv_ptr=d_uinv2.ptr, rects=bad)
    plan.undistort_image_dev(d_img.ptr, d_u.ptr, d_rec2.ptr, uinv_ptr=d_uinv2.ptr, rects=(shape[0] - 10, shape[1] - 10, 50, 50))
    plan.sync()
    assert np.array_equal(d_rec2.get(), rec_host)
    # the Lawler-Fujita scratch is part of the plan's reported workspace and does not grow from call to call
    assert plan.workspace_bytes == before > 6 * u[0].nbytes
    # against the oracle (scipy.ndimage.map_coordinates, the reference's calls)
    if dtype is np.float64:      # (f32: a sample a few ulps from the 'constant' border may land on its other side, see above)
        ref = orc.undistort_image(deformed, u)
        assert np.abs(rec_host - ref).max() < 1e-9 * np.abs(ref).max()
    plan.close()


def test_async_driver_and_plan_reuse():
    """gpa_extract_displacement_field_async on two plans (device pointers via torch), results equal
    to the synchronous host-pointer call; repeated calls with the same k-lists reuse the staged
    tables; changing the k-lists re-stages them."""
    shape = (256, 256)
    kvecs = hex_kvecs(0.1, 7.0)
    img = hex_moire(shape, kvecs, gaussian_bump_displacement(shape), noise=0.1, seed=8)
    kw, sigma, _ = orc.derive_params(kvecs)
    kl_a = np.stack(explicit_klists(kvecs, kw, 2, 2))
    kl_b = np.stack(explicit_klists(kvecs, kw, 3, 3))
    ref = {}
    plan = _lib.Plan(shape, 27, np.float64)
    for name, kl in (('a', kl_a), ('b', kl_b)):
        ref[name] = plan.extract_displacement_field(img, kvecs, kl, sigma, 2 * sigma)[0]
    plans = [_lib.Plan(shape, 27, np.float64) for _ in range(2)]
    d_img = DeviceArray(img)
    outs = [DeviceArray(np.zeros((2,) + shape)) for _ in range(2)]
    for rnd, (name, kl) in enumerate((('a', kl_a), ('a', kl_a), ('b', kl_b), ('a', kl_a))):
        for j in range(2):
            plans[j].extract_displacement_field_async(d_img.ptr, kvecs, kl, sigma, 2 * sigma, 10, outs[j].ptr)
        for j in range(2):
            plans[j].sync()
            assert plans[j].last_iters() == (10, 10)
            assert np.array_equal(outs[j].get(), ref[name]), (rnd, j)
    for p in plans + [plan]:
        p.close()


def test_error_paths_and_extreme_shapes():
    """argument checking at the C ABI and very oblong images"""
    with pytest.raises(_lib.GPAError):
        _lib.Plan((2, 64), 1, np.float32)                 # axis shorter than 4
    with pytest.raises(_lib.GPAError):
        _lib.Plan((64, 70000), 1, np.float32)             # axis longer than 65536
    # an axis too long for an LDS-resident transform (round 6): the plan exists, for the plain-DFT rows and the per-pixel
    # kernels; its sweep and unwrap entry points refuse with the reason
    big = _lib.Plan((64, 20000), 1, np.float32)
    with pytest.raises(_lib.GPAError, match='too large for the sweep'):
        big.lockin_batch(np.zeros((64, 20000), dtype=np.float32), np.zeros((1, 2)), 5.0)
    with pytest.raises(_lib.GPAError, match='too large for the sweep'):
        big.unwrap(np.zeros((64, 20000), dtype=np.float32))
    big.close()
    plan = _lib.Plan((64, 64), 2, np.float32)
    img = np.zeros((64, 64), dtype=np.float32)
    with pytest.raises(_lib.GPAError):
        plan.lockin_batch(img, np.zeros((3, 2)), 5.0)     # more k-vectors than max_batch
    with pytest.raises(_lib.GPAError):
        plan.lockin_batch(img, np.zeros((1, 2)), -1.0)    # sigma must be positive
    with pytest.raises(ValueError):
        plan.lockin_batch(np.zeros((32, 64)), np.zeros((1, 2)), 5.0)
    # an all-zero image: every |sf| is 0, nothing ever wins (kidx = -1, lock-in 0) as in the reference
    lock, kidx, _ = plan.sweep(img, (0.1, 0.0), np.array([[0.1, 0.0], [0.12, 0.0]]), 5.0)
    assert np.all(kidx == -1) and np.all(lock == 0)
    plan.close()
    # 16384 x 64 (f32) and 8192 x 100 (f64, padded second axis) against the oracle
    for shape, dt, tol in (((16384, 64), np.float32, 5e-6), ((8192, 100), np.float64, 1e-10)):
        rng = np.random.default_rng(1)
        im = rng.normal(size=shape)
        p2 = _lib.Plan(shape, 1, dt)
        out = p2.lockin_batch(im, [(0.05, 0.11)], 6.0)[0]
        assert rel(out, orc.lockin(im, (0.05, 0.11), 6.0, workers=8)) < tol
        p2.close()


# ---- f-2: phase gradient -> Jacobian -> lattice properties -------------------------------------
def _pdiff(x, y, period):
    return (np.asarray(x) - y + period / 2) % period - period / 2


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', DTYPES)
def test_f2_jacobian_and_props_golden(golden, dtype):
    from pygpa_amd import property_extract as pe
    g = golden('props_64')
    f32 = dtype == np.float32
    J = pe.phasegradient2J(g['kvecs'], g['grads'], g['weights'], 0.5, iso_ref=False, dtype=dtype)
    assert J.shape == g['J'].shape and J.dtype == dtype
    scale = np.abs(g['J']).max()
    assert np.abs(J - g['J']).max() <= (2e-5 if f32 else 1e-11) * scale
    J_iso = pe.phasegradient2J(g['aniks'], g['grads'], g['weights'], 0.5, dtype=dtype)      # iso_ref=True default
    assert np.abs(J_iso - g['J_iso']).max() <= (2e-5 if f32 else 1e-11) * np.abs(g['J_iso']).max()
    # properties of the reference's own J: angles in degrees, alpha, kappa
    for props, ref in ((pe.props_from_J(g['J'], dtype=dtype), g['props']),
                       (pe.props_from_Jac(np.eye(2) + g['J'], 3.0, 2.0, True, dtype=dtype), g['props_diff']),
                       (pe.props_from_Jac(g['jac_rand'], dtype=dtype), g['props_rand'])):
        assert props.shape == ref.shape
        kap = ref[3]
        # the anisotropy direction is ill-conditioned as kappa -> 1 (error ~ eps / (kappa - 1))
        tol_ani = (2e-5 if f32 else 1e-11) * (1 + 1 / (kap - 1)) * 57.3
        assert np.all(np.abs(_pdiff(props[0], ref[0], 360)) <= (1e-4 if f32 else 1e-10))
        assert np.all(np.abs(_pdiff(props[1], ref[1], 180)) <= tol_ani)
        assert np.allclose(props[2], ref[2], rtol=1e-5 if f32 else 1e-12, atol=(2e-6 if f32 else 1e-13) * np.abs(ref[2] * kap).max())
        assert np.allclose(props[3], ref[3], rtol=(2e-5 if f32 else 1e-11) * kap.max())


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', DTYPES)
def test_f2_props_vs_oracle_random(dtype):
    """props_from_jac against the oracle's LAPACK SVD on random Jacobians of both orientations
    (det > 0 and det < 0), away from the degenerate kappa = 1 and singular cases."""
    rng = np.random.default_rng(5)
    jac = rng.normal(size=(300, 200, 2, 2))
    ref = orc.props_from_jac(jac)
    ok = (ref[3] > 1.05) & (ref[3] < 1e3)
    props = _lib.props_from_jac(jac, dtype=dtype)
    f32 = dtype == np.float32
    assert (np.linalg.det(jac)[ok] < 0).any() and (np.linalg.det(jac)[ok] > 0).any()
    assert np.abs(_pdiff(props[0], ref[0], 360))[ok].max() <= (0.05 if f32 else 1e-8)
    assert np.abs(_pdiff(props[1], ref[1], 180))[ok].max() <= (0.02 if f32 else 1e-8)
    assert np.allclose(props[2][ok], ref[2][ok], rtol=2e-3 if f32 else 1e-9)
    assert np.allclose(props[3][ok], ref[3][ok], rtol=2e-3 if f32 else 1e-9)


@pytest.mark.gpu
def test_f2_fused_from_sweep_vs_oracle():
    """calc_props_from_phasegradient on the device sweep's own gradients against the oracle chain."""
    from pygpa_amd import geometric_phase_analysis as GPA, property_extract as pe
    n = 192
    kvecs = hex_kvecs(0.11, 4.0)
    img = hex_moire((n, n), kvecs, gaussian_bump_displacement((n, n)))
    img0 = img - img.mean()
    gs = [GPA.wfr2_grad_opt(img0, 12, k[0], k[1], kw=0.02, kstep=0.01) for k in kvecs]
    grads = np.stack([x['grad'] for x in gs])
    weights = np.stack([np.abs(x['lockin']) for x in gs])
    props = pe.calc_props_from_phasegradient(kvecs, grads, weights, nmperpixel=1.0)
    J_ref = orc.phasegradient2J(kvecs, grads, weights, 1.0, iso_ref=True)
    _, theta_0, _ = pe.get_initial_props(kvecs)
    ref = orc.props_from_jac(np.eye(2) + J_ref, refangle=theta_0)
    assert np.abs(_pdiff(props[0], ref[0], 360)).max() < 1e-9
    assert np.allclose(props[2], ref[2], rtol=1e-11) and np.allclose(props[3], ref[3], rtol=1e-10)
    well = ref[3] > 1.001
    assert np.abs(_pdiff(props[1], ref[1], 180))[well].max() < 1e-6


@pytest.mark.gpu
def test_tiled_explicit_pow2_windows_vs_oracle():
    """Power-of-two windows (native FFT lengths) around derived tiles, device-resident path against the
    oracle on the same tiling."""
    import test_distributed as TD
    from pygpa_amd import distributed as D
    img, kvecs, klists = TD._case()
    u_ref = D.extract_displacement_field_tiled(img, kvecs, klists=klists, halo=20, window=(64, 128), compute=TD._oracle_compute())
    u = D.extract_displacement_field_tiled(img, kvecs, klists=klists, halo=20, window=(64, 128))
    assert rel(u, u_ref) < 1e-8
    u32 = D.extract_displacement_field_tiled(img, kvecs, klists=klists, halo=20, window=(64, 128), dtype=np.float32)
    assert rel(u32, u_ref) < 5e-4


@pytest.mark.gpu
def test_tiled_device_resident_equals_host_staged():
    """The device-resident tile pipeline (2-D device copies into the stitched fields) and the
    host-staged one (`compute=default_compute`) run the same kernels on the same windows."""
    from pygpa_amd import distributed as D
    kvecs = hex_kvecs(0.1, 7.0)
    img = hex_moire((192, 256), kvecs, gaussian_bump_displacement((192, 256)), noise=0.05, seed=3)
    klists = explicit_klists(kvecs, 0.04, 3, 3)
    for dtype in DTYPES:
        u_dev = D.extract_displacement_field_tiled(img, kvecs, (2, 2), klists=klists, halo=24, dtype=dtype)
        comp = D.default_compute((96 + 48, 128 + 48), img.shape, 27, dtype, 0)
        u_host = D.extract_displacement_field_tiled(img, kvecs, (2, 2), klists=klists, halo=24, dtype=dtype, compute=comp)
        # the whole-image mean is reduced on the device in one path and by NumPy in the other
        assert np.abs(u_dev - u_host).max() <= (5e-4 if dtype == np.float32 else 1e-9)


# (the torch-tensor pipeline of N > 1 ranks: tests/test_gpu_configs.py::test_tiled_two_ranks_one_gpu)


# ---- f-4: Huber plane fit ------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize('dtype', DTYPES)
def test_f4_fit_plane_vs_scipy(dtype):
    """gpa_fit_plane (IRLS, device reductions) against the reference's scipy least_squares(loss='huber')
    on a noisy tilted phase map with an outlier patch and a bump; SciPy stops at ftol = 1e-8."""
    from pygpa_amd import mathtools
    rng = np.random.default_rng(0)
    n0, n1 = 200, 240
    x, y = np.meshgrid(np.arange(n0), np.arange(n1), indexing='ij')
    img = 0.013 * x - 0.021 * y + 0.4 + 0.3 * rng.normal(size=(n0, n1))
    img[50:90, 60:120] += 6.0
    img += 3 * np.exp(-((x - 150) ** 2 + (y - 50) ** 2) / 300.)
    ref = orc.fit_plane(img)
    got = mathtools.fit_plane(img.astype(dtype))
    # slopes in rad/pixel, offset in rad; f32 only rounds the input image
    assert np.allclose(got[:2], ref[:2], rtol=0, atol=1e-6 * np.abs(ref[:2]).max() + (1e-8 if dtype == np.float32 else 0))
    assert abs(got[2] - ref[2]) < (1e-5 if dtype == np.float32 else 1e-6)
    # an exact plane is recovered exactly, in one or two passes
    coef, iters = _lib.get_plan((n0, n1), 1, np.float64).fit_plane(0.5 * x - 0.25 * y + 3.0)
    assert np.allclose(coef, [0.5, -0.25, 3.0], rtol=1e-12, atol=1e-10) and iters <= 60


# ---- f-3: peak finding ---------------------------------------------------------------------------
PEAK_CASES = ['clean128', 'noisy200x240', 'weak96', 'harmonics256', 'aniso160', 'stripe128']


@pytest.mark.gpu
@pytest.mark.parametrize('name', PEAK_CASES)
def test_f3_extract_primary_ks_golden(golden, name, capsys):
    """mirror extract_primary_ks (device spectrum / smoothing / local maxima) against the reference's
    driver output: the k-vectors are grid frequencies, so they must be identical"""
    import pygpa_amd.geometric_phase_analysis as GPA
    g = golden('peaks')
    thr, dog = g[name + '_kw']
    for dtype in DTYPES:
        pks, aks = GPA.extract_primary_ks(g[name + '_image'], threshold=float(thr), DoG=bool(dog), dtype=dtype)
        assert np.array_equal(pks, g[name + '_primary']), dtype
        assert np.array_equal(aks, g[name + '_all']), dtype


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('shape,sigma,dog', [((200, 240), 1.0, 50.0), ((96, 130), 2.5, 0.0), ((64, 64), 0.7, 50.0)])
def test_f3_smoothed_spectrum_and_candidates(shape, sigma, dog, dtype):
    """smooth (incl. radius-200 kernel reflected several times over a 64-pixel axis) and the full
    peak_local_max candidate list against the oracle (SciPy's gaussian_filter / maximum_filter)"""
    kvecs = hex_kvecs(0.12, 17.0)
    img = hex_moire(shape, kvecs, noise=0.5, seed=6)
    plan = _lib.get_plan(shape, 1, dtype)
    coords, vals, smooth = plan.find_peaks(img, sigma, dog, 0.05, want_smooth=True)
    ref = orc.smoothed_spectrum(img, sigma, DoG=dog > 0)
    tol = 2e-5 if dtype == np.float32 else 1e-11
    assert rel(smooth, ref) < tol
    ref_c = orc.peak_local_max(ref, 0.05)
    if dtype == np.float64:
        # |p_hat(k)| = |p_hat(-k)| up to rounding: the order inside such a pair is not defined
        assert {tuple(c) for c in coords} == {tuple(c) for c in ref_c}
        assert np.allclose(vals, ref[tuple(ref_c.T)], rtol=1e-10)
    else:
        # f32 rounding may move candidates at the threshold / on near-ties: the strong ones must agree
        strong = ref[tuple(ref_c.T)] > 0.1 * ref.max()
        got = {tuple(c) for c in coords}
        assert all(tuple(c) in got for c in ref_c[strong])
    assert np.all(np.diff(vals) <= 0)


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', DTYPES)
def test_f4_gaussian_deconvolve_golden(golden, dtype):
    """gaussian_deconvolve against the reference's output (its padding / kernel code over the restated
    skimage Wiener filter): 94 x 115 and 110 x 131 padded shapes -> Bluestein 2-D DFTs"""
    import pygpa_amd.geometric_phase_analysis as GPA
    g = golden('deconv')
    tol = 3e-5 if dtype == np.float32 else 1e-10
    dec = GPA.gaussian_deconvolve(g['data'], float(g['sigma']), dr=int(g['dr']), balance=float(g['balance']), dtype=dtype)
    assert dec.shape == g['dec'].shape
    assert rel(dec, g['dec']) < tol
    dec_b = GPA.gaussian_deconvolve(g['data'][0], 5.0, dr=10, balance=200, dtype=dtype)
    assert rel(dec_b, g['dec_b']) < tol


# (BASELINE configs at their full sizes: tests/test_gpu_configs.py)


@pytest.mark.gpu
def test_pinned_host_arrays():
    """page-locked NumPy arrays (pinned_empty) as input and as out=: same numbers as pageable ones"""
    import gc
    import pygpa_amd
    kvecs = hex_kvecs(0.1, 7.0)
    img = hex_moire((128, 256), kvecs, gaussian_bump_displacement((128, 256)), noise=0.1, seed=3, dtype=np.float32)
    klists = np.stack(explicit_klists(kvecs, 0.04, 2, 2))
    plan = _lib.get_plan((128, 256), 12, np.float32)
    u = plan.extract_displacement_field(img, kvecs, klists, 10, 20, 10)[0]
    pimg = pygpa_amd.pinned_empty(img.shape, np.float32)
    pimg[...] = img
    pu = pygpa_amd.pinned_empty(u.shape, np.float32)
    got = plan.extract_displacement_field(pimg, kvecs, klists, 10, 20, 10, out=pu)[0]
    assert got is pu and np.array_equal(pu, u)
    with pytest.raises(ValueError):
        plan.extract_displacement_field(pimg, kvecs, klists, 10, 20, 10, out=pu[:, :, ::2])
    view = pu[1, 5:9]
    del pu, got, pimg
    gc.collect()
    assert np.isfinite(view).all()      # a view keeps the pinned allocation alive


# ---- randomized shapes / parameters ----------------------------------------------------------------
def _random_cases(seed, count):
    rng = np.random.default_rng(seed)
    sizes = [64, 96, 100, 128, 130, 200, 256, 257, 320]
    done = 0
    while done < count:
        n0, n1 = (int(v) for v in rng.choice(sizes, 2))
        # the reference's swapped-axis eigenvalue table (phase_unwrap.py:107-109) has a zero away from DC
        # once one side is at least twice the other (cos(pi I / M) = 1 at I = 2M): its own output is NaN there
        if max(n0, n1) >= 1.75 * min(n0, n1):
            continue
        done += 1
        r_k = float(rng.uniform(0.07, 0.2))
        xi = float(rng.uniform(0.0, 60.0))
        nx, ny = int(rng.integers(1, 4)), int(rng.integers(1, 4))
        sigma = int(rng.integers(3, 10))   # (the reference slices its mask with 2*sigma: integers only)
        yield (n0, n1), r_k, xi, nx, ny, sigma, int(rng.integers(0, 2 ** 31))


@pytest.mark.gpu
def test_random_shapes_driver_vs_oracle():
    """16 seeded random cases -- power-of-two, even, odd and prime-ish axis lengths mixed freely
    (padded lock-ins, Bluestein unwraps), random lattice / sigma / candidate grids -- whole fused driver in
    f64 against the oracle, and the f32 build within its stated tolerance."""
    # GPA_TEST_RANDOM_CASES / GPA_TEST_RANDOM_SEED widen the sweep for soak runs
    ncases = int(os.environ.get('GPA_TEST_RANDOM_CASES', '16'))
    for shape, r_k, xi, nx, ny, sigma, seed in _random_cases(int(os.environ.get('GPA_TEST_RANDOM_SEED', '2026')), ncases):
        kvecs = hex_kvecs(r_k, xi)
        img = hex_moire(shape, kvecs, 0.4 * gaussian_bump_displacement(shape), noise=0.2, seed=seed % 1000)
        kw = np.linalg.norm(kvecs, axis=1).mean() / 2.5
        klists = np.stack(explicit_klists(kvecs, kw, nx, ny))
        border = 2 * int(sigma)
        u_ref, parts = orc.extract_displacement_field(img, kvecs, sigma=sigma, klists=klists, return_parts=True)
        ref_kidx = np.stack([g['kidx'] for g in parts['gs']])
        tag = (shape, round(r_k, 3), round(xi, 1), nx, ny, sigma)
        plan = _lib.get_plan(shape, 3 * nx * ny, np.float64)
        u, _, kidx, iters = plan.extract_displacement_field(img, kvecs, klists, sigma, border, want_kidx=True)
        for p in range(3):   # bit-exact winners, except where two candidates tie to rounding
            check_kidx(kidx[p], ref_kidx[p], img - img.mean(), klists[p], sigma, 1e-9)
        assert rel(u, u_ref) < 1e-7, tag
        plan32 = _lib.get_plan(shape, 3 * nx * ny, np.float32)
        u32 = plan32.extract_displacement_field(img, kvecs, klists, sigma, border)[0]
        # f32: a candidate near-tie may resolve differently at isolated pixels (different lock-in there, a
        # local bump of a few tenths of a pixel after the unwrap): bound the bulk tightly, the outliers loosely
        scale = np.abs(u_ref).max()
        d32 = np.abs(u32 - u_ref) / scale
        # (ten unconverged f32 iterations may also leave a small common offset per component)
        dm = (u32 - u32.mean(axis=(1, 2), keepdims=True)) - (u_ref - u_ref.mean(axis=(1, 2), keepdims=True))
        assert np.sqrt((dm ** 2).mean()) / scale < 3e-3 and np.sqrt((d32 ** 2).mean()) < 5e-3, tag
        assert np.quantile(d32, 0.999) < 1e-2 and d32.max() < 5e-2, tag


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(256, 64), (96, 257)])
def test_unwrap_elongated_images(shape):
    """aspect ratio >= 2: the reference's swapped-axis preconditioner table has a zero eigenvalue and the
    reference returns NaN; the device path uses the true eigenvalues (oracle with compat=False)"""
    rng = np.random.default_rng(4)
    n0, n1 = shape
    x, y = np.meshgrid(np.arange(n0), np.arange(n1), indexing='ij')
    phi_true = 0.21 * x + 0.13 * y + 2.0 * np.sin(x / 17.0) * np.cos(y / 11.0)
    psi = orc.wrap_to_pi(phi_true + 0.05 * rng.normal(size=shape))
    weight = 0.5 + rng.random(shape)
    dx, dy = np.diff(psi, axis=1), np.diff(psi, axis=0)
    with np.errstate(all='ignore'):
        assert not np.isfinite(orc.unwrap_prediff(dx, dy, weight, kmax=20, compat=True)).all()
    ref = orc.unwrap_prediff(dx, dy, weight, kmax=20, compat=False)
    got, iters = _lib.get_plan(shape, 1, np.float64).unwrap_prediff(dx, dy, weight, kmax=20)
    assert np.isfinite(got).all()
    assert rel(got, ref) < 1e-8


@pytest.mark.gpu
def test_random_shapes_unwrap_and_warp_vs_oracle():
    """seeded random shapes (4 .. 150 pixels a side, any parity, aspect ratio < 2) through the unwrap
    family (wrapped phase / pre-differenced, weighted / unweighted) and the Lawler-Fujita resampling,
    f64 against the oracle"""
    import pygpa_amd.phase_unwrap as PU
    import pygpa_amd.geometric_phase_analysis as GPA
    rng = np.random.default_rng(int(os.environ.get('GPA_TEST_RANDOM_SEED', '77')))
    done = 0
    while done < int(os.environ.get('GPA_TEST_RANDOM_CASES', '14')):
        n0, n1 = (int(v) for v in rng.integers(4, 151, 2))   # (plans need at least 4 pixels a side)
        # (aspect ratio 2 is where the reference's swapped-axis table turns singular; close to it the table is
        # ill-conditioned and rounding differences are amplified, 5e-6 seen at 115 x 58)
        if max(n0, n1) >= 1.75 * min(n0, n1):
            continue
        done += 1
        shape = (n0, n1)
        x, y = np.meshgrid(np.arange(n0), np.arange(n1), indexing='ij')
        phi = rng.uniform(-0.4, 0.4) * x + rng.uniform(-0.4, 0.4) * y + 1.5 * np.sin(x / 9.0 + y / 13.0)
        psi = orc.wrap_to_pi(phi + 0.03 * rng.normal(size=shape))
        weight = None if done % 2 else 0.3 + rng.random(shape)
        kmax = int(rng.integers(3, 25))
        ref = orc.unwrap(psi, weight=weight, kmax=kmax)
        got = PU.phase_unwrap(psi, weight=weight, kmax=kmax)
        assert rel(got, ref) < 1e-8, (shape, kmax)
        got2 = PU.phase_unwrap_prediff(np.diff(psi, axis=1), np.diff(psi, axis=0), weight=weight, kmax=kmax)
        assert rel(got2, ref) < 1e-8, (shape, kmax)
        if min(n0, n1) >= 24:
            u = np.stack([1.5 * np.sin(x / 19.0) * np.cos(y / 23.0), 1.2 * np.cos(x / 17.0 + y / 29.0)])
            img = np.cos(0.5 * x) + np.sin(0.37 * y)
            inv_ref = orc.invert_u_overlap(u, iters=12)
            inv = GPA.invert_u_overlap(u, iters=12)
            assert rel(inv, inv_ref) < 1e-9, shape
            und_ref = orc.undistort_image(img, u)
            und = GPA.undistort_image(img, u)
            # mode='constant' jumps to cval where the sampling point leaves [0, n-1]: on the outermost pixel
            # ring, where u vanishes, that is decided by rounding noise of u_inv -- compare inside it
            assert rel(und[1:-1, 1:-1], und_ref[1:-1, 1:-1]) < 1e-9, shape


@pytest.mark.gpu
def test_random_shapes_spectral_helpers_vs_oracle():
    """seeded random shapes through the a9 / f-3 / f-4 entry points (arbitrary-size DFTs, Gaussian
    smoothing with reflected kernels longer than the image, Wiener deconvolution, Huber plane fit)"""
    import pygpa_amd.geometric_phase_analysis as GPA
    from pygpa_amd import mathtools
    rng = np.random.default_rng(int(os.environ.get('GPA_TEST_RANDOM_SEED', '31')))
    for _ in range(int(os.environ.get('GPA_TEST_RANDOM_CASES', '8'))):
        n0, n1 = (int(v) for v in rng.integers(20, 180, 2))
        shape = (n0, n1)
        kvecs = hex_kvecs(float(rng.uniform(0.08, 0.2)), float(rng.uniform(0, 60)))
        img = hex_moire(shape, kvecs, noise=0.3, seed=int(rng.integers(0, 1000)))
        plan = _lib.get_plan(shape, 1, np.float64)
        # a9
        assert rel(plan.per_dft(img), orc.per(img, inverse_dft=False)[0]) < 1e-10, shape
        # f-3: smoothed spectrum and candidates
        sigma = float(rng.uniform(0.6, 3.0))
        dog = bool(rng.integers(0, 2))
        coords, vals, smooth = plan.find_peaks(img, sigma, 50.0 if dog else 0.0, 0.1, want_smooth=True)
        ref = orc.smoothed_spectrum(img, sigma, DoG=dog)
        assert rel(smooth, ref) < 1e-10, (shape, sigma, dog)
        assert {tuple(c) for c in coords} == {tuple(c) for c in orc.peak_local_max(ref, 0.1)}, (shape, sigma, dog)
        # f-4: deconvolution of a smooth field, plane fit of a tilted noisy one
        dr = int(rng.integers(1, max(2, min(n0, n1) // 5)))
        x, y = np.meshgrid(np.arange(n0), np.arange(n1), indexing='ij')
        field = np.sin(x / 11.0) * np.cos(y / 7.0) + 0.01 * rng.normal(size=shape)
        s2 = float(rng.uniform(1.0, 4.0))
        assert rel(GPA.gaussian_deconvolve(field, s2, dr=dr, balance=300.0), orc.gaussian_deconvolve(field, s2, dr=dr, balance=300.0)) < 1e-9, (shape, dr)
        tilted = 0.02 * x - 0.015 * y + 0.3 + 0.2 * rng.normal(size=shape)
        tilted[: n0 // 4, : n1 // 4] += 5.0
        got, refp = mathtools.fit_plane(tilted), orc.fit_plane(tilted)
        assert np.allclose(got[:2], refp[:2], rtol=0, atol=2e-6 * np.abs(refp[:2]).max()) and abs(got[2] - refp[2]) < 2e-5, shape


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', DTYPES)
def test_ksharded_device_equals_fused_driver(dtype):
    """(peak x k-vector) sharding with the device kernels, 4 simulated ranks in one process: the gathered
    selection gives the winners and lock-ins of the single-GPU sweep (identical up to amplitude ties) and
    the same displacement field"""
    from pygpa_amd import distributed as D
    shape = (192, 256)
    kvecs = hex_kvecs(0.11, 9.0)
    img = hex_moire(shape, kvecs, gaussian_bump_displacement(shape), noise=0.2, seed=11)
    kw = np.linalg.norm(kvecs, axis=1).mean() / 2.5
    klists = np.stack(explicit_klists(kvecs, kw, 4, 4))
    u, lock, kidx = D.extract_displacement_field_ksharded(img, kvecs, sigma=9, klists=list(klists), dtype=dtype, _simulate_world=4)
    plan = _lib.get_plan(shape, 48, dtype)
    u1, lock1, kidx1, _ = plan.extract_displacement_field(img, kvecs, klists, 9, 18, want_lockins=True, want_kidx=True)
    same = kidx == kidx1
    assert same.mean() > 0.9999
    # (the fused driver takes the image mean on the device, the sharded path on the host: last-bit differences)
    assert rel(lock[same], lock1[same]) < (2e-6 if dtype == np.float32 else 1e-12)
    assert rel(u, u1) < (2e-3 if dtype == np.float32 else 1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('shape', [(256, 512), (1024, 512), (128, 512)])
def test_fused_driver_rectangular_pow2(shape, dtype):
    """rectangular power-of-two images (aspect ratio >= 2, where the reference's preconditioner table is
    singular): the fused driver -- fused unwrap kernels on unequal axis lengths -- against the oracle's
    pieces with the true eigenvalues (compat=False)"""
    kvecs = hex_kvecs(0.1, 7.0)
    img = hex_moire(shape, kvecs, gaussian_bump_displacement(shape), noise=0.1, seed=9)
    kw, sigma, _ = orc.derive_params(kvecs)
    klists = np.stack(explicit_klists(kvecs, kw, 2, 2))
    plan = _lib.get_plan(shape, 12, dtype)
    u, lock, kidx, iters = plan.extract_displacement_field(img, kvecs, klists, sigma, 2 * sigma, want_lockins=True, want_kidx=True)
    gs = [orc.sweep(img - img.mean(), sigma, klists[p], kvecs[p], workers=8) for p in range(3)]
    phases, weights, _ = orc.phases_weights(np.stack([g['lockin'] for g in gs]), sigma)
    dudx, dudy = orc.reconstruct_gradients(kvecs, phases, weights)
    wn = np.linalg.norm(weights, axis=0)
    u_ref = np.stack([orc.unwrap_prediff(dudx[c], dudy[c], wn, kmax=10, compat=False, workers=8) for c in range(2)])
    assert np.isfinite(u).all()
    if dtype == np.float64:
        assert rel(u, u_ref) < 1e-8
    else:
        d = (u - u.mean(axis=(1, 2), keepdims=True)) - (u_ref - u_ref.mean(axis=(1, 2), keepdims=True))
        assert np.sqrt((d ** 2).mean()) / np.abs(u_ref).max() < 3e-3


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', DTYPES)
def test_fused_unwrap_iteration_counts(dtype):
    """fused power-of-two unwrap at iteration counts around the search-direction ring (10): 1, 2, 3, 9, 10,
    11, 19, 20, 21, 35 -- the deferred phi updates must match the oracle's per-iteration updates"""
    rng = np.random.default_rng(8)
    shape = (128, 128)
    x, y = np.meshgrid(np.arange(128), np.arange(128), indexing='ij')
    psi = orc.wrap_to_pi(0.3 * x - 0.2 * y + 3.0 * np.sin(x / 15.0) * np.cos(y / 21.0) + 0.2 * rng.normal(size=shape))
    weight = 0.2 + rng.random(shape)
    plan = _lib.get_plan(shape, 1, dtype)
    for kmax in (1, 2, 3, 9, 10, 11, 19, 20, 21, 35):
        ref, it_ref = orc.unwrap(psi, weight=weight, kmax=kmax, return_iters=True)
        got, it = plan.unwrap(psi, weight, kmax=kmax)
        # (the f32 build stops at its residual floor, here after 31 iterations)
        assert it_ref == kmax and (it == kmax or (dtype == np.float32 and it >= 25)), (kmax, it)
        assert rel(got, ref) < (5e-4 if dtype == np.float32 else 1e-9), kmax


@pytest.mark.gpu
def test_random_shapes_lockin_sweep_grad_vs_oracle():
    """seeded random shapes / sigma / candidate lists through the sweep entry points (a1-a4): batched lock-ins,
    best-of-K with winners bit-exact up to ties, phase gradient of the winner; f64 against the oracle"""
    rng = np.random.default_rng(int(os.environ.get('GPA_TEST_RANDOM_SEED', '404')))
    for _ in range(int(os.environ.get('GPA_TEST_RANDOM_CASES', '12'))):
        n0, n1 = (int(v) for v in rng.integers(16, 300, 2))
        shape = (n0, n1)
        r_k = float(rng.uniform(0.06, 0.22))
        kvecs = hex_kvecs(r_k, float(rng.uniform(0, 60)))
        img = hex_moire(shape, kvecs, 0.3 * gaussian_bump_displacement(shape), noise=0.3, seed=int(rng.integers(0, 1000)))
        img0 = img - img.mean()
        sigma = float(rng.uniform(2.0, 9.0))
        K = int(rng.integers(1, 12))
        klist = kvecs[0] + rng.uniform(-0.4, 0.4, (K, 2)) * r_k
        plan = _lib.get_plan(shape, max(K, 3), np.float64)
        tag = (shape, round(sigma, 2), K)
        assert rel(plan.lockin_batch(img0, kvecs, sigma), orc.lockin_batch(img0, kvecs, sigma)) < 1e-11, tag
        ref = orc.sweep(img0, sigma, klist, kvecs[0], want_grad=True)
        lock, kidx, grad = plan.sweep(img0, kvecs[0], klist, sigma, want_grad=True)
        check_kidx(kidx, ref['kidx'], img0, klist, sigma, 1e-9)
        same = kidx == ref['kidx']
        assert rel(lock[same], ref['lockin'][same]) < 1e-11, tag
        # the gradient uses np.gradient of the winner's phase, defined modulo the pi-periodic wrap of
        # wrapToPi(2 g) / 2; compare where the lock-in is not vanishing
        gref = orc.wrap_to_pi(2 * ref['grad']) / 2
        d = orc.wrap_to_pi(2 * (grad - gref)) / 2
        amp = np.abs(ref['lockin'])
        ok = same & (amp > 1e-3 * amp.max())
        assert np.abs(d[ok]).max() < 1e-8, tag


# ---- the remaining spellings of rows a3 / a4 / a6 / f-1 against the reference's outputs (variants_64) --------------
@pytest.mark.gpu
@pytest.mark.parametrize('dtype', DTYPES)
def test_variants_wfr4(golden, dtype):
    """wfr4's gated selection (ordered 5 x 5 list; a generate_klists ring of 210 candidates)"""
    import pygpa_amd.geometric_phase_analysis as GPA
    g = golden('variants_64')
    img0 = g['image'] - g['image'].mean()
    sigma = int(g['sigma'])
    for klist, kref, dk, key in ((g['wfr4_klist'], g['kvecs'][0], float(g['wfr4_dk']), 'wfr4'),
                                 (g['wfr4_ring'], g['kvecs'][1], 0.005, 'wfr4_ring')):
        r = GPA.wfr4(img0, sigma, klist, kref, dk, dtype=dtype)
        same = np.all(r['w'] == g[key + '_w'], axis=0)
        if dtype is np.float64:
            assert same.all()
        else:
            assert same.mean() > 0.97        # f32 amplitude near-ties flip a gated chain
        assert rel(r['lockin'][same], g[key + '_lockin'][same]) < TOL[dtype]['lock']
    # generate_klists is host bookkeeping: the list the golden ring was made from
    ring = GPA.generate_klists(g['kvecs'], kmax=1.12, kmin=0.9, sort_list=True)[1]
    assert np.array_equal(ring, g['wfr4_ring'])


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', DTYPES)
def test_variants_gradient_spellings(golden, dtype):
    """wfr2_grad (None / 'diff' / callable), cuGPA 'diff' / callable, *_vec aliases, against the reference's outputs"""
    import pygpa_amd.cuGPA as cuGPA
    import pygpa_amd.geometric_phase_analysis as GPA
    g = golden('variants_64')
    h = golden('hex_64')
    img0 = g['image'] - g['image'].mean()
    sigma, pk, kw, kstep = int(g['sigma']), g['kvecs'][0], float(g['kw']), float(g['kstep'])
    tol = TOL[dtype]['grad'] * np.pi

    def close(a, ref, kidx):
        ok = kidx == h['a3_kidx'][0]
        assert ok.mean() > (0.999 if dtype is np.float64 else 0.99)
        assert np.array_equal(np.isnan(a), np.isnan(ref))
        d = np.abs(orc.wrap_to_pi(2 * (a - ref)) / 2)[ok]        # results live on a circle of circumference pi
        assert np.nanmax(d) < tol

    r = GPA.wfr2_grad(img0, sigma, pk[0], pk[1], kw, kstep, dtype=dtype)
    close(r['grad'], g['wfr2_grad_none'], r['kidx'])
    r = GPA.wfr2_grad(img0, sigma, pk[0], pk[1], kw, kstep, grad='diff', dtype=dtype)
    close(r['grad'], g['wfr2_grad_diff'], r['kidx'])
    rc = GPA.wfr2_grad(img0, sigma, pk[0], pk[1], kw, kstep, dtype=dtype,
                       grad=lambda ph: np.stack([np.diff(ph, axis=1, append=np.nan), np.diff(ph, axis=0, append=np.nan)], axis=-1))
    close(rc['grad'], g['wfr2_grad_diff'], rc['kidx'])
    rv = GPA.wfr2_grad_vec(img0, sigma, pk[0], pk[1], kw, kstep, dtype=dtype)
    ro = GPA.wfr2_grad_opt(img0, sigma, pk[0], pk[1], kw, kstep, dtype=dtype)
    assert np.array_equal(rv['grad'], ro['grad']) and np.array_equal(rv['lockin'], ro['lockin'])
    assert np.array_equal(GPA.wfr2_only_lockin_vec(img0, sigma, pk[0], pk[1], kw, kstep, dtype=dtype), ro['lockin'])
    if dtype is np.float64:
        c = cuGPA.wfr2_grad_opt(img0, sigma, pk[0], pk[1], kw, kstep, grad='diff')
        kidx = GPA.optwfr2(img0, sigma, pk[0], pk[1], kw, kstep)['kidx']
        close(c['grad'], g['cu_grad_diff'], kidx)
        c2 = cuGPA.wfr2_grad_opt(img0, sigma, pk[0], pk[1], kw, kstep,
                                 grad=lambda ph: (np.diff(ph, axis=0, append=np.nan), np.diff(ph, axis=1, append=np.nan)))
        close(c2['grad'], g['cu_grad_diff'], kidx)
        assert np.array_equal(cuGPA.wfr2_only_grad(img0, sigma, pk, kw, kstep, grad='diff'), c['grad'], equal_nan=True)
    else:
        cs = cuGPA.wfr2_grad_single(img0, sigma, pk[0], pk[1], kw, kstep)
        assert 'w' not in cs
        close(cs['grad'], g['cu_single_grad'], GPA.optwfr2(img0, sigma, pk[0], pk[1], kw, kstep, dtype=dtype)['kidx'])


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', DTYPES)
def test_variants_invert_u_and_prediff(golden, dtype):
    import pygpa_amd.geometric_phase_analysis as GPA
    g = golden('variants_64')
    tol = 1e-10 if dtype is np.float64 else 2e-4
    for key, kw in (('invert_u', dict()), ('invert_u_edge2_it5', dict(iters=5, edge=2))):
        out = GPA.invert_u(-g['warp_u'], dtype=dtype, **kw)
        assert out.shape == g[key].shape
        assert np.abs(out - g[key]).max() < tol * max(1.0, np.abs(g[key]).max())
    for key, wu in (('u_prediff', True), ('u_prediff_unweighted', False)):
        u = GPA.reconstruct_u_inv_from_phases(g['kvecs'], g['prediff_grads'], g['prediff_weights'], weighted_unwrap=wu,
                                              pre_diff=True, dtype=dtype)
        assert rel(u, g[key]) < (1e-8 if dtype is np.float64 else 5e-4), key


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('n', [64, 256, 1024, 2048])
def test_transform_free_column_solve_equals_dct_solve(n, dtype):
    """square power-of-two images solve the columns of the preconditioner without a transform
    (colsolve_tri_kernel: recursive filters + scans); GPA_COLSOLVE=tri / fft selects the column kernel (default: tri for f64, DCT for f32).  Same iterates up to rounding: iteration counts equal, phi within the PCG tolerance."""
    rng = np.random.default_rng(n)
    kvecs = hex_kvecs(0.1, 7.0)
    img = hex_moire((n, n), kvecs, gaussian_bump_displacement((n, n)), noise=0.2, seed=n)
    klists = np.stack(explicit_klists(kvecs, 0.04, 2, 2))
    # a harder unwrap: rough weights, noisy phase, 40 iterations
    w = rng.uniform(0.05, 1.0, size=(n, n))
    psi = orc.wrap_to_pi(0.05 * np.add.outer(np.arange(n) ** 1.1, np.arange(n)) + rng.normal(scale=0.3, size=(n, n)))
    plan = _lib.Plan((n, n), 12, dtype)
    out = {}
    for mode in ('tri', 'fft'):
        _lib.set_option('COLSOLVE', mode)
        try:
            out[mode] = plan.extract_displacement_field(img, kvecs, klists, 10, 20, kmax=10)
            out[mode + '_u'] = plan.unwrap(psi, w, kmax=40)
        finally:
            _lib.set_option('COLSOLVE', None)
    plan.close()
    tol = 1e-9 if dtype is np.float64 else 3e-5
    assert out['tri'][3] == out['fft'][3]
    assert rel(out['tri'][0], out['fft'][0]) < tol
    assert rel(out['tri_u'][0], out['fft_u'][0]) < tol
    if dtype is np.float64:
        assert out['tri_u'][1] == out['fft_u'][1]


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('shape,knx', [((512, 512), 4), ((200, 300), 3), ((1024, 256), 2), ((96, 64), 5)])
def test_candidate_split_pass_b_is_bit_identical(shape, knx, dtype):
    """small images run the K candidates of a row on up to four workgroups and merge the partial winners
    (launch_passB_split); GPA_NO_KSPLIT=1 (read at plan creation) keeps them in one workgroup: same bits"""
    kvecs = hex_kvecs(0.1, 7.0)
    img = hex_moire(shape, kvecs, gaussian_bump_displacement(shape), noise=0.3, seed=21)
    klists = np.stack(explicit_klists(kvecs, 0.04, knx, knx))
    K = knx * knx
    _lib.set_option('NO_KSPLIT', '1')
    ref_plan = _lib.Plan(shape, 3 * K, dtype)
    _lib.set_option('NO_KSPLIT', None)
    plan = _lib.Plan(shape, 3 * K, dtype)
    a = ref_plan.extract_displacement_field(img, kvecs, klists, 10, 20, want_lockins=True, want_kidx=True)
    b = plan.extract_displacement_field(img, kvecs, klists, 10, 20, want_lockins=True, want_kidx=True)
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[1], b[1]) and np.array_equal(a[0], b[0])
    img0 = img - img.mean()
    s1 = ref_plan.sweep(img0, kvecs[1], klists[1], 10)
    s2 = plan.sweep(img0, kvecs[1], klists[1], 10)
    assert np.array_equal(s1[0], s2[0]) and np.array_equal(s1[1], s2[1])
    ref_plan.close()
    plan.close()


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('shape', [(300, 200), (1100, 600), (130, 1030)])
def test_compact_padded_axes(shape, dtype, monkeypatch, gpa_option):
    """non-power-of-two axes, narrow spatial kernel: the plan switches to the compact extension (lags -E .. E,
    transform length pow2 >= n + 2E instead of >= 2n - 1) -- same numbers as the oracle and as the full
    extension, and switching sigma on one plan (compact -> full -> compact) restages every table"""
    kvecs = hex_kvecs(0.11, 4.0)
    img = hex_moire(shape, kvecs, gaussian_bump_displacement(shape), noise=0.2, seed=5)
    img0 = img - img.mean()
    klist = explicit_klists(kvecs, 0.03, 3, 3)[0]
    plan = _lib.Plan(shape, len(klist), dtype)
    full_len = [plan.fft_len(0), plan.fft_len(1)]
    gpa_option('NO_COMPACT', '1')
    plan_full = _lib.Plan(shape, len(klist), dtype)
    gpa_option('NO_COMPACT', None)
    for sigma in (4, 60, 4, 9):
        ref = orc.sweep(img0, sigma, klist, kvecs[0], workers=8)
        lock, kidx, _ = plan.sweep(img0, kvecs[0], klist, sigma)
        lens = [plan.fft_len(0), plan.fft_len(1)]
        if sigma == 4:     # E ~ 31 (f64) | 24 (f32): 300 -> 512 (full 1024), 1100 -> 2048 (full 4096), 1030 -> 2048 (full 4096)
            assert lens[0] < full_len[0] or lens[1] < full_len[1], (lens, full_len)
        if sigma == 60 and shape == (300, 200):    # kernel as wide as the image: back to the full extension
            assert lens == full_len
        check_kidx(kidx, ref['kidx'], img0, klist, sigma, TOL[dtype]['tie'])
        same = kidx == ref['kidx']
        assert same.mean() > 0.999
        assert rel(lock[same], ref['lockin'][same]) < TOL[dtype]['lock'], (sigma, lens)
        lock_f, kidx_f, _ = plan_full.sweep(img0, kvecs[0], klist, sigma)
        assert [plan_full.fft_len(0), plan_full.fft_len(1)] == full_len
        same = kidx == kidx_f
        assert same.mean() > 0.999 and rel(lock[same], lock_f[same]) < TOL[dtype]['lock']
        # the unfused lock-in entry point shares the tables
        lb = plan.lockin_batch(img0, klist[:2], sigma)
        assert rel(lb, orc.lockin_batch(img0, klist[:2], sigma)) < TOL[dtype]['lock']
    plan.close()
    plan_full.close()


def _unwrap_case(shape, seed, weighted=True):
    rng = np.random.default_rng(seed)
    n0, n1 = shape
    x, y = np.meshgrid(np.arange(n0), np.arange(n1), indexing='ij')
    phi = 0.23 * x - 0.17 * y + 2.5 * np.sin(x / (0.11 * n0 + 3.0)) * np.cos(y / (0.07 * n1 + 5.0))
    psi = orc.wrap_to_pi(phi + 0.05 * rng.normal(size=shape))
    weight = 0.3 + rng.random(shape) if weighted else None
    return psi, weight


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(48, 80), (100, 60), (300, 200), (500, 500), (130, 104), (360, 364), (1000, 1500),
                                   (66, 88), (512, 384), (36, 36), (100, 100), (1000, 1000), (1200, 1200),
                                   (63, 65), (250, 250), (126, 90), (1001, 1001), (75, 77)])
def test_mixed_radix_fused_unwrap_vs_oracle(shape, monkeypatch, gpa_option):
    """image sizes that factor into 2, 3, 5, 7, 11, 13 run the fused 4-kernel PCG on the mixed-radix FFT (square ones
    with the transform-free column solve on ragged chunks; rows that are not whole 4-pixel vectors -- 63 x 65, 250^2,
    1001^2 -- through the one-pixel instantiations of the stencil, flush and row kernels):
    same numbers as the oracle (f64 1e-8, f32 within the PCG tolerance), the same iteration count as the
    Bluestein path (GPA_NO_MR=1) and as the oracle's loop"""
    psi, weight = _unwrap_case(shape, 31 + shape[0])
    for kmax in (7, 40):
        ref, ref_iters = orc.unwrap(psi, weight=weight, kmax=kmax, return_iters=True)
        plan = _lib.Plan(shape, 1, np.float64)
        got, iters = plan.unwrap(psi, weight, kmax=kmax)
        assert rel(got, ref) < 1e-8, (shape, kmax)
        assert iters == ref_iters, (shape, kmax, iters, ref_iters)
        gpa_option('NO_MR', '1')
        plan_b = _lib.Plan(shape, 1, np.float64)
        gpa_option('NO_MR', None)
        got_b, iters_b = plan_b.unwrap(psi, weight, kmax=kmax)
        assert iters_b == iters and rel(got, got_b) < 1e-9
        if shape[0] == shape[1]:
            # square: the columns are solved without a transform by default; the mixed-radix column kernel must agree
            gpa_option('COLSOLVE', 'fft')
            got_f, iters_f = plan.unwrap(psi, weight, kmax=kmax)
            gpa_option('COLSOLVE', None)
            assert iters_f == iters and rel(got, got_f) < 1e-9
        plan32 = _lib.Plan(shape, 1, np.float32)
        got32, _ = plan32.unwrap(psi.astype(np.float32), None if weight is None else weight.astype(np.float32), kmax=kmax)
        assert rel(got32 - got32.mean(), ref - ref.mean()) < 2e-4, (shape, kmax)
        for p in (plan, plan_b, plan32):
            p.close()


def _smooth_lengths(lo, hi):
    out = []
    for n in range(lo, hi + 1):
        m = n
        for r in (2, 3, 5, 7, 11, 13):
            while m % r == 0:
                m //= r
        if m == 1:
            out.append(n)
    return out


@pytest.mark.gpu
def test_random_smooth_shapes_unwrap_vs_oracle():
    """seeded random shapes whose sides factor into 2..13 (any mix of radices, square and rectangular, rows a
    multiple of 4 or not): unwrap family in f64 against the oracle, iteration counts included.
    GPA_TEST_RANDOM_CASES / GPA_TEST_RANDOM_SEED widen the sweep for soak runs."""
    rng = np.random.default_rng(int(os.environ.get('GPA_TEST_RANDOM_SEED', '99')))
    lengths = _smooth_lengths(8, 720)
    done, want = 0, int(os.environ.get('GPA_TEST_RANDOM_CASES', '12'))
    while done < want:
        n0 = int(rng.choice(lengths))
        n1 = n0 if rng.random() < 0.4 else int(rng.choice(lengths))
        if max(n0, n1) >= 1.75 * min(n0, n1):
            continue
        done += 1
        shape = (n0, n1)
        psi, weight = _unwrap_case(shape, int(rng.integers(0, 2 ** 31)), weighted=bool(done % 3))
        kmax = int(rng.integers(2, 30))
        ref, ref_iters = orc.unwrap(psi, weight=weight, kmax=kmax, return_iters=True)
        plan = _lib.get_plan(shape, 1, np.float64)
        got, iters = plan.unwrap(psi, weight, kmax=kmax)
        assert rel(got, ref) < 1e-8, (shape, kmax, rel(got, ref))
        assert iters == ref_iters, (shape, kmax, iters, ref_iters)
        got2, _ = plan.unwrap_prediff(np.diff(psi, axis=1), np.diff(psi, axis=0), weight, kmax=kmax)
        assert rel(got2, ref) < 1e-8, (shape, kmax)


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(68, 68), (136, 116), (204, 236), (332, 332), (1392, 1040), (1006, 1004), (97, 101), (202, 202)])
def test_chirpz_fused_unwrap_vs_oracle(shape, monkeypatch, gpa_option):
    """sides with a prime factor > 13 (17, 29, 59, 83, 251, 503): the fused iteration with chirp-z DFTs on a smooth
    L >= 2n - 1 of the mixed-radix engine (transform-free columns when square) against the oracle and against
    the power-of-two Bluestein path (GPA_NO_MR=1)"""
    psi, weight = _unwrap_case(shape, 7 + shape[1])
    for kmax in (6, 30):
        ref, ref_iters = orc.unwrap(psi, weight=weight, kmax=kmax, return_iters=True)
        plan = _lib.Plan(shape, 1, np.float64)
        got, iters = plan.unwrap(psi, weight, kmax=kmax)
        assert rel(got, ref) < 1e-8 and iters == ref_iters, (shape, kmax, rel(got, ref), iters, ref_iters)
        gpa_option('NO_MR', '1')
        plan_b = _lib.Plan(shape, 1, np.float64)
        gpa_option('NO_MR', None)
        got_b, iters_b = plan_b.unwrap(psi, weight, kmax=kmax)
        assert iters_b == iters and rel(got, got_b) < 1e-9
        if shape[0] == shape[1]:
            gpa_option('COLSOLVE', 'fft')   # the chirp-z column kernel instead of the recursions
            got_f, iters_f = plan.unwrap(psi, weight, kmax=kmax)
            gpa_option('COLSOLVE', None)
            assert iters_f == iters and rel(got, got_f) < 1e-9
        plan32 = _lib.Plan(shape, 1, np.float32)
        got32, _ = plan32.unwrap(psi.astype(np.float32), weight.astype(np.float32), kmax=kmax)
        assert rel(got32 - got32.mean(), ref - ref.mean()) < 2e-4, (shape, kmax)
        for p in (plan, plan_b, plan32):
            p.close()


@pytest.mark.gpu
def test_chirpz_forced_on_smooth_lengths(monkeypatch, gpa_option):
    """GPA_MR_FORCE_BLUESTEIN=1: the chirp-z route on lengths that have a direct plan must give the direct route's
    iterates"""
    shape = (300, 200)
    psi, weight = _unwrap_case(shape, 3)
    plan = _lib.Plan(shape, 1, np.float64)
    gpa_option('MR_FORCE_BLUESTEIN', '1')
    plan_z = _lib.Plan(shape, 1, np.float64)
    gpa_option('MR_FORCE_BLUESTEIN', None)
    a, ia = plan.unwrap(psi, weight, kmax=25)
    b, ib = plan_z.unwrap(psi, weight, kmax=25)
    assert ia == ib and rel(a, b) < 1e-9
    plan.close()
    plan_z.close()


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('shape', [(128, 128), (256, 512), (100, 60), (300, 300), (136, 116), (63, 65), (250, 250),
                                   (64, 2048), (72, 3000), (1024, 128), (512, 96)])
def test_image_stack_equals_single_images(shape, dtype, monkeypatch, gpa_option):
    """gpa_extract_displacement_field_batch_dev: a stack of images through one set of unwrap launches
    (blockIdx.z = problem) -- every image's u and iteration counts equal the single-image driver's bit for bit
    (power-of-two, smooth, square smooth with transform-free columns, chirp-z sizes; rows that take the shared-forward
    pass B, periodic and zero-padded; columns of 1024 / 512 points, whose column kernel takes 4 pairs per workgroup).  Bit for bit with the same
    kernels on both sides (GPA_NO_LAT=1: a single small power-of-two image otherwise runs latency-tuned
    instantiations of the row / column kernels, whose multiply-adds the compiler contracts differently); with the
    default kernels: to rounding, 2e-5 (f32) / 1e-12 (f64) of max |u|, and the same iteration counts."""
    gpa_option('NO_LAT', '1')
    kvecs = hex_kvecs(0.12, 5.0)
    sigma = 5
    klists = np.stack(explicit_klists(kvecs, 0.04, 2, 2))
    B = 5
    imgs = np.stack([hex_moire(shape, kvecs, (0.3 + 0.2 * i) * gaussian_bump_displacement(shape), noise=0.05 * (i + 1), seed=i)
                     for i in range(B)])
    imgs[3] = imgs[3, ::-1, ::-1]   # (different images converge after different iteration counts)
    plan = _lib.Plan(shape, 12, dtype)
    u_b, it_b = plan.extract_displacement_field_stack(imgs, kvecs, klists, sigma, 2 * sigma, kmax=10)
    for i in range(B):
        u, _, _, iters = plan.extract_displacement_field(imgs[i], kvecs, klists, sigma, 2 * sigma, kmax=10)
        assert np.array_equal(u_b[i], u), (shape, i, float(np.abs(u_b[i] - u).max()))
        assert tuple(it_b[i]) == tuple(iters), (shape, i)
    # a second call with another stack size reallocates the batch workspace
    u_b2, _ = plan.extract_displacement_field_stack(imgs[:2], kvecs, klists, sigma, 2 * sigma, kmax=10)
    assert np.array_equal(u_b2, u_b[:2])
    # chunks of 2 frames (2 + 2 + 1) through the double-buffered upload / compute / download pipeline
    u_c, it_c = plan.extract_displacement_field_stack(imgs, kvecs, klists, sigma, 2 * sigma, kmax=10, chunk=2)
    assert np.array_equal(u_c, u_b) and np.array_equal(it_c, it_b)
    # default kernels for the single image
    gpa_option('NO_LAT', None)
    # (2e-5: the 64 x 2048 frames reach 1.01e-5)
    tol = (2e-5 if dtype == np.float32 else 1e-12) * float(np.abs(u_b).max())
    for i in range(B):
        u, _, _, iters = plan.extract_displacement_field(imgs[i], kvecs, klists, sigma, 2 * sigma, kmax=10)
        assert float(np.abs(u_b[i] - u).max()) <= tol, (shape, i, float(np.abs(u_b[i] - u).max()), tol)
        assert tuple(it_b[i]) == tuple(iters), (shape, i)
    plan.close()


@pytest.mark.gpu
def test_random_stacks_equal_single_images(monkeypatch, gpa_option):
    """(GPA_NO_LAT=1: the same kernels for a stack and for a single image, see test_image_stack_equals_single_images)
    seeded random (shape, stack size, precision) draws -- any parity of the sides, power-of-two, smooth and chirp-z
    lengths mixed -- the stack call equals the single-image driver frame for frame, bit for bit.
    GPA_TEST_RANDOM_CASES / GPA_TEST_RANDOM_SEED widen the sweep."""
    gpa_option('NO_LAT', '1')
    rng = np.random.default_rng(int(os.environ.get('GPA_TEST_RANDOM_SEED', '5')))
    want = max(4, int(os.environ.get('GPA_TEST_RANDOM_CASES', '10')) // 2)
    sizes = [32, 48, 60, 63, 64, 65, 68, 75, 96, 100, 116, 128, 130, 160, 250, 256]
    done = 0
    while done < want:
        n0, n1 = (int(v) for v in rng.choice(sizes, 2))
        if max(n0, n1) >= 1.75 * min(n0, n1):
            continue
        done += 1
        shape = (n0, n1)
        dtype = np.float64 if rng.random() < 0.5 else np.float32
        B = int(rng.integers(1, 7))
        kvecs = hex_kvecs(float(rng.uniform(0.1, 0.2)), float(rng.uniform(0, 60)))
        sigma = int(rng.integers(3, 7))
        klists = np.stack(explicit_klists(kvecs, 0.03, int(rng.integers(1, 3)), int(rng.integers(1, 3))))
        imgs = np.stack([hex_moire(shape, kvecs, 0.3 * gaussian_bump_displacement(shape), noise=0.05 + 0.1 * i, seed=int(rng.integers(0, 1000)))
                         for i in range(B)])
        plan = _lib.get_plan(shape, klists.shape[0] * klists.shape[1], dtype)
        u_b, it_b = plan.extract_displacement_field_stack(imgs, kvecs, klists, sigma, 2 * sigma, kmax=10, chunk=int(rng.integers(1, B + 1)))
        for i in range(B):
            u, _, _, iters = plan.extract_displacement_field(imgs[i], kvecs, klists, sigma, 2 * sigma, kmax=10, want_kidx=True)
            assert np.array_equal(u_b[i], u), (shape, np.dtype(dtype).name, B, i)
            assert tuple(it_b[i]) == tuple(iters), (shape, B, i)


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('n', [256, 512, 1024])
def test_latency_tuned_kernels_equal_lean_kernels(n, dtype, monkeypatch, gpa_option):
    """One image up to 1024^2 runs latency-tuned instantiations of the fused PCG kernels (every input requested before
    the first wait; pq_small_kernel instead of the sliding-window stencil); stacks and larger images run the lean ones.
    Same formulas: the driver's u agrees to rounding (1e-5 / 1e-12 of max |u|) with the same iteration counts, and both
    hold the oracle's bar of test_driver_vs_oracle_512 indirectly through it."""
    shape = (n, n)
    kvecs = hex_kvecs(0.1, 7.0)
    sigma = 6
    klists = np.stack(explicit_klists(kvecs, 0.03, 2, 2))
    img = hex_moire(shape, kvecs, 0.5 * gaussian_bump_displacement(shape), noise=0.1, seed=3)
    plan = _lib.Plan(shape, 12, dtype)
    u_lat, _, _, it_lat = plan.extract_displacement_field(img, kvecs, klists, sigma, 2 * sigma, kmax=10)
    gpa_option('NO_LAT', '1')
    u_lean, _, _, it_lean = plan.extract_displacement_field(img, kvecs, klists, sigma, 2 * sigma, kmax=10)
    gpa_option('NO_LAT', None)
    assert tuple(it_lat) == tuple(it_lean)
    tol = (1e-5 if dtype == np.float32 else 1e-12) * float(np.abs(u_lean).max())
    assert float(np.abs(u_lat - u_lean).max()) <= tol
    plan.close()
