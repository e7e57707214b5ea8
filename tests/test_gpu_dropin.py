"""The reference's own hot-path tests, restated against the drop-in modules
(pygpa_amd.geometric_phase_analysis / cuGPA / phase_unwrap).  GPU only."""
import numpy as np
import pytest

import pygpa_amd.cuGPA as cuGPA
import pygpa_amd.geometric_phase_analysis as GPA
import pygpa_amd.phase_unwrap as pu
from pygpa_amd.synthetic import hex_kvecs, gaussian_bump_displacement, hex_moire

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def testset():
    shape = (512, 512)
    ks = hex_kvecs(0.1, 7.0)
    u = gaussian_bump_displacement(shape)
    original = hex_moire(shape, ks)
    deformed = hex_moire(shape, ks, u)
    noise = hex_moire(shape, ks, None, noise=1.0, seed=0) - original
    return original, deformed, noise, ks, u


def test_displacement_field(testset):
    """reference tests/test_geometric_phase_analysis.py:61-66 and tests/test_cuGPA.py:46-52"""
    original, deformed, noise, ks, u_true = testset
    for kw in (dict(), dict(wfr_func=GPA.wfr2_grad_opt)):
        u = -GPA.extract_displacement_field(deformed + noise, ks, **kw)
        assert u.shape == u_true.shape
        assert np.all(np.abs(u - u_true)[:, 20:-20, 20:-20] < 0.9)
    u32 = -GPA.extract_displacement_field(deformed + noise, ks, dtype=np.float32)
    assert np.all(np.abs(u32 - u_true)[:, 20:-20, 20:-20] < 0.9)
    # :66-69: the noise-free image with the Wiener deconvolution of the lock-in Gaussian
    u2 = -GPA.extract_displacement_field(deformed, ks, deconvolve=True)
    assert u2.shape == u_true.shape
    assert np.all(np.abs(u2 - u_true)[:, 20:-20, 20:-20] < 0.05)
    u1 = -GPA.extract_displacement_field(deformed, ks)
    assert np.abs(u2 - u_true)[:, 20:-20, 20:-20].max() < np.abs(u1 - u_true)[:, 20:-20, 20:-20].max()


def test_displacement_field_stack(testset, monkeypatch, gpa_option):
    """the stack form (one device call for all frames) returns what a loop over the reference-shaped
    extract_displacement_field returns, frame by frame, to the last bit (same kernels on both sides: GPA_NO_LAT,
    see test_image_stack_equals_single_images) -- and passes the reference's own bar"""
    gpa_option('NO_LAT', '1')
    original, deformed, noise, ks, u_true = testset
    frames = np.stack([deformed + noise, deformed, deformed[::-1, ::-1] + 0.5 * noise])
    for dtype in (np.float64, np.float32):
        us = GPA.extract_displacement_field_stack(frames, ks, dtype=dtype)
        assert us.shape == (3,) + u_true.shape
        for i in range(3):
            assert np.array_equal(us[i], GPA.extract_displacement_field(frames[i], ks, dtype=dtype))
        assert np.all(np.abs(-us[0] - u_true)[:, 20:-20, 20:-20] < 0.9)


@pytest.mark.parametrize('wfr_func', [cuGPA.wfr2_grad_opt, cuGPA.wfr2_grad_single])
def test_cugpa_displacement_field(testset, wfr_func):
    """reference tests/test_cuGPA.py:45-56, literally: the cuGPA module's sweeps plugged in as `wfr_func`
    (the 'any other callable' branch of extract_displacement_field: positional (image, sigma, kx, ky) +
    kw= / kstep= keywords, 'lockin' read from the returned mapping)"""
    original, deformed, noise, ori_ks, gaussiandeform = testset
    u = -GPA.extract_displacement_field(deformed + noise, ori_ks[:3], wfr_func=wfr_func)
    assert u.shape == gaussiandeform.shape
    assert np.all(np.abs(u - gaussiandeform)[:, 20:-20, 20:-20] < 0.9)
    u2 = -GPA.extract_displacement_field(deformed, ori_ks[:3], deconvolve=True)
    assert u2.shape == gaussiandeform.shape
    assert np.all(np.abs(u2 - gaussiandeform)[:, 20:-20, 20:-20] < 0.05)
    # the plug-in branch and the fused branch are the same numbers (f64 wrapper) / f32-close (single wrapper)
    u_fused = -GPA.extract_displacement_field(deformed + noise, ori_ks[:3])
    tol = 1e-9 if wfr_func is cuGPA.wfr2_grad_opt else 2e-3
    assert np.abs(u - u_fused).max() < tol * np.abs(u_fused).max()


@pytest.mark.parametrize('wfr_func1,wfr_func2', [(GPA.optwfr2, cuGPA.wfr2_grad_opt), (GPA.wfr2_grad_opt, cuGPA.wfr2_grad_opt)])
def test_cugpa_wfr2_variants_lockin(wfr_func1, wfr_func2, testset):
    """reference tests/test_cuGPA.py:66-82, literally"""
    original, deformed, noise, ori_ks, _ = testset
    kw = np.linalg.norm(ori_ks, axis=1).mean() / 2.5
    sigma = int(np.ceil(1 / np.linalg.norm(ori_ks, axis=1).min()))
    kstep = kw / 3
    gs = [wfr_func1(deformed - deformed.mean(), sigma, pk[0], pk[1], kw=kw, kstep=kstep) for pk in ori_ks]
    gs2 = [wfr_func2(deformed - deformed.mean(), sigma, pk[0], pk[1], kw=kw, kstep=kstep) for pk in ori_ks]
    for g1, g2 in zip(gs, gs2):
        assert np.allclose(g1['lockin'], g2['lockin'])


def test_displacement_field_500():
    """the reference's own image size (500^2, not a power of two): padded-mode lock-ins + Bluestein unwrap"""
    shape = (500, 500)
    ks = hex_kvecs(0.1, 7.0)
    u_true = gaussian_bump_displacement(shape)
    deformed = hex_moire(shape, ks, u_true, noise=0.5, seed=4)
    u = -GPA.extract_displacement_field(deformed, ks)
    assert np.all(np.abs(u - u_true)[:, 20:-20, 20:-20] < 0.9)
    from oracle import gpa_oracle as orc
    u_ref = -orc.extract_displacement_field(deformed, ks, workers=8)
    assert np.abs(u - u_ref).max() < 1e-7 * np.abs(u_ref).max()


def test_wfr2_variants_lockin(testset):
    """reference tests/test_geometric_phase_analysis.py:82-97 and tests/test_cuGPA.py:68-82"""
    original, deformed, noise, ks, _ = testset
    kw = np.linalg.norm(ks, axis=1).mean() / 2.5
    sigma = int(np.ceil(1 / np.linalg.norm(ks, axis=1).min()))
    kstep = kw / 3
    img0 = deformed - deformed.mean()
    for f1, f2 in [(GPA.optwfr2, GPA.wfr2), (GPA.optwfr2, cuGPA.wfr2_grad_opt), (GPA.wfr2_grad_opt, cuGPA.wfr2_grad_opt)]:
        for pk in ks:
            g1 = f1(img0, sigma, pk[0], pk[1], kw=kw, kstep=kstep)
            g2 = f2(img0, sigma, pk[0], pk[1], kw=kw, kstep=kstep)
            assert np.allclose(g1['lockin'], g2['lockin'])
    g = cuGPA.wfr2_only_lockin(img0, sigma, ks[0], kw, kstep)
    assert np.allclose(g, GPA.optwfr2(img0, sigma, ks[0][0], ks[0][1], kw, kstep)['lockin'])
    gs = cuGPA.wfr2_grad_single(img0, sigma, ks[0][0], ks[0][1], kw, kstep)
    assert 'w' not in gs and gs['lockin'].dtype == np.complex64


@pytest.mark.parametrize('kmax', [1, 7, 30])
def test_equivalent_phase_unwrap_variants(kmax):
    """reference tests/test_phase_unwrap.py:11-31, :49-75 (N = 256 linear ramp)"""
    N = 256
    xx, yy = np.meshgrid(np.arange(N), np.arange(N), indexing='ij')
    psi0 = (yy + xx) / (4 * np.sqrt(2))
    psi = pu._wrapToPi(psi0)
    weight = np.ones_like(psi)
    res = pu.phase_unwrap_ref(psi=psi, weight=weight, kmax=kmax)
    assert np.allclose(res - res.mean(), psi0 - psi0.mean())
    assert np.allclose(res, pu.phase_unwrap(psi=psi, weight=weight, kmax=kmax))
    assert np.allclose(res, pu.phase_unwrap(psi=psi, weight=None, kmax=kmax))
    dx = np.diff(psi, axis=1)
    dy = np.diff(psi, axis=0)
    res_pd = pu.phase_unwrap_ref_prediff(dx=dx, dy=dy, weight=weight, kmax=kmax)
    assert np.allclose(res_pd - res_pd.mean(), psi0 - psi0.mean())
    assert np.allclose(res_pd, pu.phase_unwrap_prediff(dx=dx, dy=dy, weight=None, kmax=kmax))
    assert np.allclose(res_pd, res)


def test_equivalent_phase_unwrap_gaussian_weight():
    """reference tests/test_phase_unwrap.py:34-46"""
    N = 256
    xx, yy = np.meshgrid(np.arange(N), np.arange(N), indexing='ij')
    psi0 = (yy + xx) / (4 * np.sqrt(2))
    psi = pu._wrapToPi(psi0)
    gaussian = np.exp(-((xx - N // 2) ** 2 + (yy - N // 2) ** 2) / (0.3 * N ** 2))
    assert np.allclose(pu.phase_unwrap(psi=psi, weight=None), pu.phase_unwrap(psi=psi, weight=gaussian))


# ---- tests/test_property_extract.py:25-44 (test_props_from_J), seeded draws instead of hypothesis
@pytest.mark.gpu
def test_props_from_J():
    from pygpa_amd import property_extract as pe
    from pygpa_amd.mathtools import periodic_difference
    rng = np.random.default_rng(2)
    n = 4000
    theta = rng.uniform(0., 360., n)
    psi = rng.uniform(-90., 90., n)
    kappa = np.exp(rng.uniform(np.log(1. + 1e-7), np.log(1e4), n))
    a = np.exp(rng.uniform(np.log(1e-10), np.log(1e10), n))

    def rot(deg):
        r = np.deg2rad(deg)
        return np.moveaxis(np.array([[np.cos(r), -np.sin(r)], [np.sin(r), np.cos(r)]]), -1, 0)
    W, V = rot(theta), rot(psi)
    D = np.zeros((n, 2, 2))
    D[:, 0, 0] = kappa * a
    D[:, 1, 1] = a
    Jac_ori = np.swapaxes(V, -1, -2) @ D @ V @ W
    props = pe.props_from_Jac(Jac_ori)
    # the direction of the anisotropy loses digits as kappa -> 1 (the reference's atol=1e-5 holds for
    # its LAPACK path under the same conditioning: 1e-16 / 1e-7 rad)
    ani_tol = 1e-5 + 1e-13 / (kappa - 1)
    assert np.allclose(periodic_difference(props[0], theta, period=360), 0, atol=1e-6)
    assert np.all(np.abs(periodic_difference(props[1], psi, period=180)) <= ani_tol)
    assert np.allclose(props[2], a)
    assert np.allclose(props[3], kappa)
    props2 = pe.props_from_J(Jac_ori / a[:, None, None] - np.eye(2), refscale=1.0)
    assert np.allclose(periodic_difference(props2[0], theta, period=360), 0, atol=1e-6)
    assert np.all(np.abs(periodic_difference(props2[1], psi, period=180)) <= ani_tol)
    assert np.allclose(props2[2], 1.0)
    assert np.allclose(props2[3], kappa)


# ---- tests/test_geometric_phase_analysis.py:44-58 (test_extract_primary_ks), seeded draws; the lattice
# is the sum-of-cosines generator of pygpa_amd.synthetic instead of latticegen.hexlattice_gen
@pytest.mark.gpu
def test_extract_primary_ks():
    import pygpa_amd.geometric_phase_analysis as GPA
    from pygpa_amd.synthetic import hex_kvecs, hex_moire
    rng = np.random.default_rng(12)
    size = 128
    for _ in range(12):
        r_k, theta = rng.uniform(0.03, 0.24), rng.uniform(0., 60.)
        ori_ks = hex_kvecs(r_k, theta, n=6)
        original = hex_moire((size, size), ori_ks[:3])
        ext_ks, _ = GPA.extract_primary_ks(original, DoG=False)
        abs_diffs = np.linalg.norm((ext_ks[None] - ori_ks[:, None]), axis=-1).min(axis=0)
        assert np.all(abs_diffs < 1.5 / size), (r_k, theta, ext_ks)


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [np.float64, np.float32])
@pytest.mark.parametrize('name', ['hex_64', 'hex_48x80', 'hex_63x65'])
def test_myweighed_lstsq_is_part_of_the_module_surface(golden, name, dtype):
    """`myweighed_lstsq(b, K, w)` (geometric_phase_analysis.py:97-113; the reference's own property_extract.py:10 imports it
    from the module): the per-pixel weighted least squares of reconstruct_u_inv_from_phases (:234-237) called the way the
    reference calls it -- wrapped phase differences, K = 2 pi kvecs, the FULL weight stack (the difference grids are one
    pixel shorter) -- against the reference's dudx / dudy"""
    import pygpa_amd.geometric_phase_analysis as GPA
    from pygpa_amd.mathtools import wrapToPi
    g = golden(name)
    K = 2 * np.pi * g['kvecs']
    dbdx = wrapToPi(np.diff(g['a5_phases'], axis=2))
    dbdy = wrapToPi(np.diff(g['a5_phases'], axis=1))
    tol = 1e-11 if dtype is np.float64 else 2e-5
    for b, ref in ((dbdx, g['a6_dudx']), (dbdy, g['a6_dudy'])):
        out = GPA.myweighed_lstsq(b, K, g['a5_weights'], dtype=dtype)
        assert out.shape == ref.shape
        assert np.abs(out - ref).max() < tol * np.abs(ref).max()
