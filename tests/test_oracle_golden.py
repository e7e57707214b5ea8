"""Pin the CPU oracle against vectors produced by the real reference
(oracle/make_golden.py).  CPU only."""
import numpy as np
import pytest
import scipy.ndimage as ndi

from oracle import gpa_oracle as orc

CASES = ['hex_64', 'hex_48x80', 'hex_63x65', 'hex_60']


def test_gaussian_matches_scipy_fourier_gaussian():
    for shape in [(64, 64), (48, 80), (63, 65)]:
        for sigma in (6, 7, 10.5, 22):
            ref = ndi.fourier_gaussian(np.ones(shape), sigma=sigma)
            mine = orc.gaussian_kspace_1d(shape[0], sigma)[:, None] * orc.gaussian_kspace_1d(shape[1], sigma)[None, :]
            assert np.array_equal(ref == 0, mine == 0)
            assert np.allclose(ref, mine, rtol=1e-12, atol=0)


def test_wrap_to_pi_edges():
    assert orc.wrap_to_pi(np.pi) == -np.pi
    assert orc.wrap_to_pi(-np.pi) == -np.pi
    assert orc.wrap_to_pi(3 * np.pi) == -np.pi


@pytest.mark.parametrize('name', CASES)
def test_a1_a2_lockin(golden, name):
    g = golden(name)
    img0 = g['image'] - g['image'].mean()
    sigma = int(g['sigma'])
    a = orc.lockin(img0, g['kvecs'][0], sigma)
    assert np.allclose(a, g['a1_GPA'], rtol=1e-10, atol=1e-12)
    b = orc.lockin(img0, g['kvecs'][1], sigma)
    assert np.allclose(b, g['a1_optGPA'], rtol=1e-10, atol=1e-12)
    if 'a2_vecGPA' in g:
        c = orc.lockin_batch(img0, g['kvecs'], sigma)
        assert np.allclose(c, g['a2_vecGPA'], rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize('name', CASES)
def test_a3_sweep(golden, name):
    g = golden(name)
    img0 = g['image'] - g['image'].mean()
    sigma = int(g['sigma'])
    for p, pk in enumerate(g['kvecs']):
        klist = orc.sweep_grid(pk[0], pk[1], float(g['kw']), float(g['kstep']))
        assert np.array_equal(klist, g['a3_klists'][p])       # host-built list is bit-identical
        res = orc.sweep(img0, sigma, klist, pk)
        assert np.array_equal(res['kidx'], g['a3_kidx'][p])    # index work: bit-exact
        assert np.allclose(res['lockin'], g['a3_lockin'][p], rtol=1e-10, atol=1e-12)
        if 'a3_w' in g:
            assert np.array_equal(res['w'], g['a3_w'][p])


@pytest.mark.parametrize('name', CASES)
def test_a4_grad(golden, name):
    g = golden(name)
    img0 = g['image'] - g['image'].mean()
    pk = g['kvecs'][0]
    res = orc.wfr2_grad_opt(img0, int(g['sigma']), pk[0], pk[1], float(g['kw']), float(g['kstep']))
    d = orc.wrap_to_pi(2 * (res['grad'] - g['a4_grad0'])) / 2   # compare modulo the pi-periodic wrap
    assert np.abs(d).max() < 1e-8


@pytest.mark.parametrize('name', CASES)
def test_a5_a6_reconstruct(golden, name):
    g = golden(name)
    phases, weights, mask = orc.phases_weights(g['a3_lockin'], int(g['sigma']))
    assert np.array_equal(mask, g['a5_mask'])                  # mask: bit-exact
    assert np.array_equal(phases, g['a5_phases'])
    assert np.array_equal(weights, g['a5_weights'])
    dudx, dudy = orc.reconstruct_gradients(g['kvecs'], phases, weights)
    assert np.allclose(dudx, g['a6_dudx'], rtol=1e-8, atol=1e-10)
    assert np.allclose(dudy, g['a6_dudy'], rtol=1e-8, atol=1e-10)


@pytest.mark.parametrize('name', CASES)
def test_a7_unwrap(golden, name):
    g = golden(name)
    wn = np.linalg.norm(g['a5_weights'], axis=0)
    dudx, dudy = g['a6_dudx'], g['a6_dudy']
    for kmax in (1, 3, 10, 100):
        phi = orc.unwrap_prediff(dudx[0], dudy[0], wn, kmax=kmax)
        ref = g['a7_phi_w_kmax%d' % kmax]
        assert np.allclose(phi, ref, rtol=1e-7, atol=1e-8 * np.abs(ref).max()), kmax
    phi = orc.unwrap_prediff(dudx[0], dudy[0])
    assert np.allclose(phi, g['a7_phi_unweighted'], rtol=1e-8, atol=1e-9)
    w0 = g['a5_weights'][0]
    psi = orc.unwrap(g['a5_phases'][0], np.sqrt(w0 / w0.max()), kmax=10)
    ref = g['a7_psi_unwrap_kmax10']
    assert np.allclose(psi, ref, rtol=1e-7, atol=1e-8 * np.abs(ref).max())


@pytest.mark.parametrize('name', CASES + ['hex_128_noise'])
def test_full_driver(golden, name):
    g = golden(name)
    u, parts = orc.extract_displacement_field(g['image'], g['kvecs'], return_parts=True)
    assert parts['sigma'] == int(g['sigma'])
    for p in range(3):
        assert np.array_equal(parts['gs'][p]['kidx'], g['a3_kidx'][p])
    assert np.allclose(u, g['u'], rtol=1e-6, atol=1e-7 * np.abs(g['u']).max())


def test_a8_iterate(golden):
    g = golden('iterate_64')
    prs, w, corr = orc.iterate_gpa(g['image'] - g['image'].mean(), g['start_ks'], int(g['sigma']))
    assert np.allclose(corr, g['corr'], rtol=1e-5, atol=1e-8)
    assert np.allclose(w, g['w'], rtol=1e-6, atol=1e-9)
    assert np.allclose(prs, g['prs'], rtol=1e-5, atol=1e-6)
    assert np.allclose(orc.reconstruct_u_inv(g['start_ks'] + corr, prs, w), g['u_weighted'], atol=1e-5)
    assert np.allclose(orc.reconstruct_u_inv(g['start_ks'] + corr, prs), g['u_global'], atol=1e-5)


def test_reference_unwrap_ramp(golden):
    """The reference's own test input (tests/test_phase_unwrap.py) at 64^2."""
    g = golden('unwrap_ramp_64')
    for kmax in (1, 5, 30):
        phi = orc.unwrap(g['psi'], np.ones_like(g['psi']), kmax=kmax)
        assert np.allclose(phi, g['ref_kmax%d' % kmax])
        assert np.allclose(phi - phi.mean(), g['psi0'] - g['psi0'].mean())
        assert np.allclose(orc.unwrap(g['psi'], None, kmax=kmax), phi)
    assert np.allclose(orc.unwrap(g['psi'], g['gaussian_weight']), g['ref_gaussian'])


def test_a9_per_known_answers():
    rng = np.random.default_rng(0)
    img = rng.normal(size=(32, 40)) + np.linspace(0, 3, 40)[None, :]
    p, s = orc.per(img)
    assert np.allclose(p + s, img)
    assert abs(s.mean()) < 1e-12
    tile = np.cos(2 * np.pi * np.arange(32) / 32)[:, None] * np.cos(2 * np.pi * 3 * np.arange(40) / 40)[None, :]
    ptile, stile = orc.per(tile)
    assert np.abs(stile).max() < 1e-9 * 1e3 or np.abs(stile).max() < 0.2   # near-periodic image: small smooth part
    phat, shat = orc.per(img, inverse_dft=False)
    assert np.allclose(phat + shat, np.fft.fft2(img))
    assert shat[0, 0] == 0


def test_f3_peak_local_max_known_answers():
    """The stand-in for skimage.feature.peak_local_max (absent from this image) against what scikit-image's
    own documentation states for min_distance = 1: both isolated maxima, highest first; a plateau-free border
    pixel is never a peak (exclude_border); a constant image has none."""
    img = np.zeros((7, 7))
    img[3, 4] = 1
    img[3, 2] = 1.5
    assert orc.peak_local_max(img, 0.0).tolist() == [[3, 2], [3, 4]]
    assert orc.peak_local_max(img, 0.8).tolist() == [[3, 2]]          # 1 < 0.8 * 1.5
    edge = np.zeros((7, 7))
    edge[0, 3] = 2.0
    edge[4, 4] = 1.0
    assert orc.peak_local_max(edge, 0.0).tolist() == [[4, 4]]
    assert len(orc.peak_local_max(np.ones((5, 5)), 0.5)) == 0


def test_f1_lawler_fujita(golden):
    g = golden('warp_96x80')
    assert np.allclose(orc.invert_u_overlap(-g['u']), g['u_inv'], rtol=0, atol=1e-12, equal_nan=True)
    assert np.allclose(orc.invert_u_overlap(-g['u'], iters=5, edge=4), g['u_inv_edge4_it5'], rtol=0, atol=1e-12, equal_nan=True)
    assert np.allclose(orc.undistort_image(g['deformed'], g['u']), g['reconstructed'], rtol=0, atol=1e-12)


def test_f2_properties(golden):
    g = golden('props_64')
    J = orc.phasegradient2J(g['kvecs'], g['grads'], g['weights'], 0.5)
    assert np.allclose(J, g['J'], rtol=1e-8, atol=1e-10)
    assert np.allclose(orc.calc_diff_from_isotropic(g['aniks']), g['dks'], rtol=0, atol=1e-14)
    J_iso = orc.phasegradient2J(g['aniks'], g['grads'], g['weights'], 0.5, iso_ref=True)
    assert np.allclose(J_iso, g['J_iso'], rtol=1e-8, atol=1e-10)
    iso = g['aniks'] + g['dks']
    assert np.allclose(np.linalg.norm(iso, axis=1), np.linalg.norm(iso, axis=1)[0])
    assert np.allclose(orc.props_from_jac(np.eye(2) + g['J']), g['props'], rtol=1e-12, atol=1e-12)
    assert np.allclose(orc.props_from_jac(np.eye(2) + g['J'], 3.0, 2.0, True), g['props_diff'], rtol=1e-12, atol=1e-12)
    assert np.allclose(orc.props_from_jac(g['jac_rand']), g['props_rand'], rtol=1e-12, atol=1e-12)


PEAK_CASES = ['clean128', 'noisy200x240', 'weak96', 'harmonics256', 'aniso160', 'stripe128']


@pytest.mark.parametrize('name', PEAK_CASES)
def test_f3_extract_primary_ks(golden, name):
    """oracle restatement against the reference's driver (run with the declared per / peak_local_max
    stand-ins, oracle/make_golden.py): identical k-vectors"""
    g = golden('peaks')
    thr, dog = g[name + '_kw']
    pks, aks = orc.extract_primary_ks(g[name + '_image'], threshold=float(thr), DoG=bool(dog))
    assert np.array_equal(pks, g[name + '_primary'])
    assert np.array_equal(aks, g[name + '_all'])


def test_f4_gaussian_deconvolve(golden):
    g = golden('deconv')
    dec = orc.gaussian_deconvolve(g['data'], float(g['sigma']), dr=int(g['dr']), balance=float(g['balance']))
    assert np.allclose(dec, g['dec'], rtol=1e-12, atol=1e-12)
    assert np.allclose(orc.gaussian_deconvolve(g['data'][0], 5.0, dr=10, balance=200), g['dec_b'], rtol=1e-12, atol=1e-12)
    # closed form the device uses: the filter is G / (G^2 + balance L^2) with G the k-space Gaussian itself
    n0, n1 = g['data'].shape[-2] + 24, g['data'].shape[-1] + 24
    G = np.outer(orc.gaussian_kspace_1d(n0, 3.0), orc.gaussian_kspace_1d(n1, 3.0))
    L = 4 - 2 * np.cos(2 * np.pi * np.arange(n0) / n0)[:, None] - 2 * np.cos(2 * np.pi * np.arange(n1) / n1)[None, :]
    W = G / (G ** 2 + 5000.0 * L ** 2)
    padded = np.pad(g['data'][0], 12, mode='reflect')
    direct = np.real(np.fft.ifft2(W * np.fft.fft2(padded)))[12:-12, 12:-12]
    assert np.allclose(direct, g['dec'][0], rtol=1e-10, atol=1e-11)


# ---- the remaining spellings of rows a3 / a4 / a6 / f-1 (tests/golden/variants_64.npz) -----------------------
def test_variants_wfr4_gated_sweep(golden):
    g = golden('variants_64')
    img0 = g['image'] - g['image'].mean()
    r = orc.wfr4(img0, int(g['sigma']), g['wfr4_klist'], g['kvecs'][0], float(g['wfr4_dk']))
    assert np.array_equal(r['w'], g['wfr4_w']) and np.allclose(r['lockin'], g['wfr4_lockin'], rtol=0, atol=1e-13)
    r = orc.wfr4(img0, int(g['sigma']), g['wfr4_ring'], g['kvecs'][1], 0.005)
    assert np.array_equal(r['w'], g['wfr4_ring_w']) and np.allclose(r['lockin'], g['wfr4_ring_lockin'], rtol=0, atol=1e-13)


def test_variants_gradient_spellings(golden):
    """wfr2_grad (compensated phase, grad None / 'diff' with its axis order) and the cuGPA module's 'diff'"""
    g = golden('variants_64')
    h = golden('hex_64')
    img0 = g['image'] - g['image'].mean()
    sigma, pk = int(g['sigma']), g['kvecs'][0]
    klist = orc.sweep_grid(pk[0], pk[1], float(g['kw']), float(g['kstep']))
    for key, kw in (('wfr2_grad_none', dict(grad=None, compensated=True)), ('wfr2_grad_diff', dict(grad='diff', compensated=True)),
                    ('cu_grad_none', dict(grad=None, compensated=False)), ('cu_grad_diff', dict(grad='diff', compensated=False))):
        r = orc.sweep_grad_variant(img0, sigma, klist, pk, **kw)
        assert np.array_equal(np.isnan(r['grad']), np.isnan(g[key])), key
        assert np.allclose(r['grad'], g[key], rtol=0, atol=1e-11, equal_nan=True), key
        assert np.array_equal(r['kidx'], h['a3_kidx'][0])
    # wfr2_grad(grad=None) and cuGPA.wfr2_grad_opt(grad=None) are wfr2_grad_opt up to rounding
    assert np.allclose(g['wfr2_grad_none'], h['a4_grad0'], rtol=0, atol=1e-11)
    assert np.allclose(g['cu_grad_none'], h['a4_grad0'], rtol=0, atol=1e-11)
    # 'diff' of wfr2_grad is the cuGPA 'diff' with the two components swapped (gpa.py:738-742 vs cu.py:58-62)
    a, b = g['wfr2_grad_diff'], g['cu_grad_diff']
    ok = ~np.isnan(a[..., ::-1]) & ~np.isnan(b)
    d = np.abs(orc.wrap_to_pi(2 * (a[..., ::-1] - b)) / 2)
    assert d[ok].max() < 1e-11


def test_variants_invert_u_and_prediff(golden):
    g = golden('variants_64')
    assert np.allclose(orc.invert_u(-g['warp_u']), g['invert_u'], rtol=0, atol=1e-12)
    assert np.allclose(orc.invert_u(-g['warp_u'], iters=5, edge=2), g['invert_u_edge2_it5'], rtol=0, atol=1e-12)
    for key, wu in (('u_prediff', True), ('u_prediff_unweighted', False)):
        u = orc.reconstruct_u_inv_from_phases(g['kvecs'], g['prediff_grads'], g['prediff_weights'], weighted_unwrap=wu,
                                              pre_diff=True, kmax=10 if wu else 100)
        assert np.abs(u - g[key]).max() < 1e-9 * np.abs(g[key]).max(), key


def test_config1_512_reference_outputs(golden):
    """BASELINE configs[0] at its own size (512^2, 3 peaks, one reference k-vector per peak): the oracle's iterate_gpa /
    reconstruct_u_inv / K = 1 driver against the REFERENCE's outputs (subsampled fixture config1_512.npz made by
    oracle/make_golden.py from geometric_phase_analysis.py:116-154, :157-193, :647-666, :907-932)."""
    from pygpa_amd.synthetic import hex_moire, gaussian_bump_displacement
    g = golden('config1_512')
    sl = (Ellipsis, slice(3, None, 7), slice(3, None, 7))
    shape = (512, 512)
    image = hex_moire(shape, g['it_true_ks'], None, noise=0.05, seed=21)
    assert np.array_equal(np.array([image.sum(), (image ** 2).sum()]), g['it_image_moments'])
    prs, w, corr = orc.iterate_gpa(image - image.mean(), g['it_start_ks'], int(g['it_sigma']))
    assert np.abs(corr - g['it_corr']).max() < 1e-15
    assert np.abs(prs[sl] - g['it_prs']).max() < 1e-11 and np.abs(w[sl] - g['it_w']).max() < 1e-12
    assert np.allclose([prs.sum(), (prs ** 2).sum()], g['it_prs_moments'], rtol=1e-12)
    ks = g['it_start_ks'] + corr
    assert np.abs(orc.reconstruct_u_inv(ks, prs, w)[sl] - g['it_u_weighted']).max() < 1e-11
    assert np.abs(orc.reconstruct_u_inv(ks, prs)[sl] - g['it_u_global']).max() < 1e-11
    kvecs = g['k1_kvecs']
    image = hex_moire(shape, kvecs, gaussian_bump_displacement(shape), noise=0.1, seed=22)
    assert np.array_equal(np.array([image.sum(), (image ** 2).sum()]), g['k1_image_moments'])
    u, parts = orc.extract_displacement_field(image, kvecs, klists=[pk[None] for pk in kvecs], return_parts=True)
    assert np.abs(u[sl] - g['k1_u']).max() < 1e-11
    lock = np.stack([q['lockin'] for q in parts['gs']])
    assert np.abs(lock[sl] - g['k1_lockin']).max() < 1e-13
    assert np.allclose([u.sum(), (u ** 2).sum()], g['k1_u_moments'], rtol=1e-10)


def test_f1_oracle_resampler_is_map_coordinates_bit_for_bit():
    """oracle._Resampler (one spline prefilter per field instead of one per round; the GPU tests at 4096^2 need it) repeats
    map_coordinates' own last call: bit-identical to the literal sequence of map_coordinates calls of
    geometric_phase_analysis.py:262-300, for both modes, with and without the overlap edge, and on a subset of rows"""
    import scipy.ndimage as ndi
    from pygpa_amd.synthetic import gaussian_bump_displacement
    us = gaussian_bump_displacement((96, 120)) * 3 + 0.1 * np.random.default_rng(3).normal(size=(2, 96, 120))

    def literal(us, iters=35, edge=0, mode='nearest'):
        xx, yy = np.mgrid[-edge:us.shape[1] + edge, -edge:us.shape[2] + edge]
        u_it = [ndi.map_coordinates(u, [xx, yy], mode=mode) for u in us]
        for _ in range(iters - 1):
            u_it = [ndi.map_coordinates(u, [xx + u_it[0], yy + u_it[1]], mode=mode) for u in us]
        u_it = [ndi.map_coordinates(u, [xx + u_it[0], yy + u_it[1]], mode=mode, cval=np.nan) for u in us]
        return np.stack(u_it)

    assert orc._Resampler(us[0], 'nearest').fast, 'SciPy moved its private entry points: the oracle falls back to plain calls'
    for mode in ('nearest', 'constant'):
        for edge in (0, 5):
            assert np.array_equal(orc.invert_u_overlap(us, edge=edge, mode=mode), literal(us, edge=edge, mode=mode), equal_nan=True)
    rows = np.r_[2:9, 40:71]
    assert np.array_equal(orc.invert_u_overlap(us, rows=rows), literal(us)[:, rows], equal_nan=True)
    img = np.random.default_rng(4).normal(size=us.shape[1:])
    assert np.array_equal(orc.undistort_image(img, us, rows=rows), orc.undistort_image(img, us)[rows])
