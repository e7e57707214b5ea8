"""BASELINE.json configs at their full sizes, the measured fp32 accuracy, the multi-rank device path and the
table-staging machinery of the fused driver.  Run with `-m gpu` on an MI355X.

Accuracy numbers are written to gpurun_out/r02_accuracy.json (copied to profiles/ when committed)."""
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import pytest

from conftest import kidx_mismatch_is_tie
import tolerances as TOL
from oracle import gpa_oracle as orc
from pygpa_amd import _lib
from pygpa_amd.synthetic import hex_kvecs, gaussian_bump_displacement, hex_moire, explicit_klists

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ACC_PATH = os.path.join(ROOT, 'gpurun_out', 'r02_accuracy.json')


def _record(key, value):
    os.makedirs(os.path.dirname(ACC_PATH), exist_ok=True)
    data = {}
    if os.path.exists(ACC_PATH):
        try:
            data = json.load(open(ACC_PATH))
        except Exception:
            data = {}
    data[key] = value
    json.dump(data, open(ACC_PATH, 'w'), indent=1, sort_keys=True)


def _px_errors(u, ref, border):
    """max / rms of |u - ref| in pixels, whole image and interior; the mean of each component is free
    (phase_unwrap.py:110-114), so it is removed first"""
    d = (u - u.mean(axis=(1, 2), keepdims=True)) - (ref - ref.mean(axis=(1, 2), keepdims=True))
    di = d[:, border:-border, border:-border]
    return {'max_px': float(np.abs(d).max()), 'rms_px': float(np.sqrt((d.astype(np.float64) ** 2).mean())),
            'interior_max_px': float(np.abs(di).max()), 'interior_rms_px': float(np.sqrt((di.astype(np.float64) ** 2).mean()))}


def _check_kidx_ties(kidx, ref_kidx, img0, klists, sigma, tie_tol):
    """every winner that differs from the reference's must be an amplitude tie (per peak)"""
    worst = 0.0
    for p in range(len(klists)):
        bad = kidx[p] != ref_kidx[p]
        worst = max(worst, float(bad.mean()))
        if bad.any():
            amps = np.abs(orc.lockin_batch(img0, klists[p], sigma, workers=os.cpu_count() or 1))
            assert np.all(kidx_mismatch_is_tie(amps, kidx[p], ref_kidx[p], tie_tol)[bad]), \
                'peak %d: %d kidx mismatches that are not amplitude ties' % (p, int(bad.sum()))
    return worst


def _set_floor(v):
    if v is None:
        _lib.set_option('F32_EPS_FLOOR', None)
    else:
        _lib.set_option('F32_EPS_FLOOR', v)


# ---- the sweep's compensation tables must survive per_dft / find_peaks / deconvolve -----
def test_sweep_tables_survive_spectral_calls():
    import pygpa_amd.geometric_phase_analysis as GPA
    shape = (128, 160)
    kvecs = hex_kvecs(0.1, 7.0)
    img = hex_moire(shape, kvecs, gaussian_bump_displacement(shape), noise=0.1, seed=4)
    img2 = hex_moire(shape, kvecs, gaussian_bump_displacement(shape), noise=0.2, seed=5)
    for dtype in (np.float64, np.float32):
        u1 = GPA.extract_displacement_field(img, kvecs, dtype=dtype)
        GPA.extract_primary_ks(img2, dtype=dtype)          # per_dft + find_peaks on the SAME cached plan
        GPA.per(img2, dtype=dtype)
        u2 = GPA.extract_displacement_field(img, kvecs, dtype=dtype)   # same k-lists: tables are not re-staged
        assert np.array_equal(u1, u2)
        plan = _lib.get_plan(shape, 64, dtype)
        klist = GPA._sweep_list(kvecs[0][0], kvecs[0][1], 0.04, 0.04 / 3)
        a = plan.sweep(img, kvecs[0], klist, 10)
        plan.per_dft(img2)
        plan.find_peaks(img2, 1.0, 50.0, 0.5)
        b = plan.sweep(img, kvecs[0], klist, 10)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    # deconvolve runs on a plan of the padded shape: share it with a sweep of that shape
    pshape = (shape[0] + 80, shape[1] + 80)
    imgp = hex_moire(pshape, kvecs, noise=0.1, seed=6)
    plan = _lib.get_plan(pshape, 64, np.float64)
    klist = GPA._sweep_list(kvecs[1][0], kvecs[1][1], 0.04, 0.04 / 3)
    a = plan.sweep(imgp, kvecs[1], klist, 10)
    GPA.gaussian_deconvolve(np.zeros((2,) + shape), 10, dr=20)
    b = plan.sweep(imgp, kvecs[1], klist, 10)
    assert np.array_equal(a[0], b[0])


def test_download_async_pipeline():
    """gpa_download_async / gpa_download_wait: u of step i lands in pinned memory while step i + 1 runs"""
    import pygpa_amd
    from test_gpu_parity import DeviceArray
    shape = (512, 512)
    kvecs = hex_kvecs(0.1, 7.0)
    klists = np.stack(explicit_klists(kvecs, 0.04, 2, 2))
    imgs = [hex_moire(shape, kvecs, gaussian_bump_displacement(shape), noise=0.1, seed=s, dtype=np.float32) for s in range(4)]
    plan = _lib.Plan(shape, 12, np.float32)
    ref = [plan.extract_displacement_field(im, kvecs, klists, 10, 20)[0] for im in imgs]
    d_imgs = [DeviceArray(im) for im in imgs]
    d_u = [DeviceArray(np.zeros((2,) + shape, dtype=np.float32)) for _ in range(2)]
    h_u = [pygpa_amd.pinned_empty((2,) + shape, np.float32) for _ in range(2)]
    got = []
    nsteps = 12
    for st in range(nsteps):
        j = st & 1
        plan.download_wait(j)                       # the copy that last read d_u[j] / wrote h_u[j] has landed
        if st >= 2:
            got.append(((st - 2) % 4, h_u[j].copy()))
        plan.extract_displacement_field_async(d_imgs[st % 4].ptr, kvecs, klists, 10, 20, 10, d_u[j].ptr)
        plan.download_async(h_u[j], d_u[j].ptr, j)
    for st in (nsteps - 2, nsteps - 1):
        plan.download_wait(st & 1)
        got.append((st % 4, h_u[st & 1].copy()))
    plan.sync()
    for i, u in got:
        assert np.array_equal(u, ref[i]), i
    plan.close()


# ---- configs[0]: 512^2, 3 Bragg peaks, a single reference k-vector per peak ---------------------------------------
C1_SL = (Ellipsis, slice(3, None, 7), slice(3, None, 7))      # oracle/make_golden.py: C1_STRIDE


def _moments(a):
    return np.array([a.sum(), (np.abs(a) ** 2).sum()])


@pytest.mark.parametrize('dtype', [np.float64, np.float32])
def test_config1_512_iterate_gpa_route(golden, dtype):
    """configs[0] on the HIP path at its own size: iterate_GPA (K = 1 lock-in per peak, weighted unwraps with kmax_iter
    25 / final kmax 200 on the 502^2 cropped maps, Huber plane fits) + reconstruct_u_inv (weighted per-pixel and
    global), geometric_phase_analysis.py:116-154, :157-193 -- against the REFERENCE's outputs (subsampled fixture
    tests/golden/config1_512.npz) and against the oracle on every pixel."""
    import pygpa_amd.geometric_phase_analysis as GPA
    g = golden('config1_512')
    shape = (512, 512)
    image = hex_moire(shape, g['it_true_ks'], None, noise=0.05, seed=21)
    assert np.array_equal(_moments(image), g['it_image_moments'])          # the fixture's image, regenerated
    start, sigma = g['it_start_ks'], int(g['it_sigma'])
    prs_o, w_o, corr_o = orc.iterate_gpa(image - image.mean(), start, sigma)
    prs, w, corr = GPA.iterate_GPA(image - image.mean(), start, sigma, dtype=dtype)
    assert prs.shape == (3, 502, 502)
    # the refined vectors land on the true ones (the reference gets within 3.3e-7)
    assert np.abs(start + corr - g['it_true_ks']).max() < 1e-6
    scale = np.abs(g['it_prs']).max()
    if dtype is np.float64:
        T = TOL.C1_F64
        assert np.abs(corr - g['it_corr']).max() < T['corr'] and np.abs(corr - corr_o).max() < T['corr']
        assert np.abs(prs[C1_SL] - g['it_prs']).max() < T['prs'] and np.abs(prs - prs_o).max() < T['prs']
        assert np.allclose(w[C1_SL], g['it_w'], rtol=1e-9, atol=1e-12) and np.allclose(w, w_o, rtol=1e-9, atol=1e-12)
        assert np.allclose(_moments(prs), g['it_prs_moments'], rtol=1e-9)
    else:
        T = TOL.C1_F32
        assert np.abs(corr - g['it_corr']).max() < T['corr']
        assert np.abs(prs[C1_SL] - g['it_prs']).max() < T['prs_rel'] * scale
        assert np.abs(prs - prs_o).max() < T['prs_rel'] * scale
        assert np.allclose(w, w_o, rtol=2e-5, atol=1e-5)
    _record('config1_512_iterate_%s' % np.dtype(dtype).name,
            {'corr_err': float(np.abs(corr - g['it_corr']).max()), 'prs_err': float(np.abs(prs - prs_o).max()),
             'prs_max': float(scale), 'k_err': float(np.abs(start + corr - g['it_true_ks']).max())})
    # reconstruct_u_inv on the reference's own phases: weighted per-pixel solve on the device, global solve on the host
    ks = start + g['it_corr']
    uw = GPA.reconstruct_u_inv(ks, prs_o, weights=w_o, dtype=dtype)
    ug = GPA.reconstruct_u_inv(ks, prs_o)
    uw_o = orc.reconstruct_u_inv(ks, prs_o, w_o)
    tol_u = TOL.C1_F64['u'] if dtype is np.float64 else 2e-5 * max(1.0, np.abs(uw_o).max())
    assert np.abs(uw - uw_o).max() < tol_u
    assert np.abs(uw[C1_SL] - g['it_u_weighted']).max() < tol_u + 1e-9
    assert np.abs(ug[C1_SL] - g['it_u_global']).max() < 1e-9


@pytest.mark.parametrize('dtype', [np.float64, np.float32])
def test_config1_512_single_kvector_driver(golden, dtype):
    """configs[0] through the driver: extract_displacement_field with ONE candidate per peak (klists=[[pk]]; the
    reference: wfr3 with klist = [pk] as wfr_func, geometric_phase_analysis.py:647-666, :907-932) at 512^2 -- the
    K = 1 instantiation of every sweep kernel -- against the reference's u / lock-ins and the oracle's on every pixel."""
    import pygpa_amd.geometric_phase_analysis as GPA
    g = golden('config1_512')
    shape = (512, 512)
    kvecs = g['k1_kvecs']
    u_true = gaussian_bump_displacement(shape)
    image = hex_moire(shape, kvecs, u_true, noise=0.1, seed=22)
    assert np.array_equal(_moments(image), g['k1_image_moments'])
    klists = [pk[None] for pk in kvecs]
    u_o, parts = orc.extract_displacement_field(image, kvecs, klists=klists, return_parts=True)
    u, gs = GPA.extract_displacement_field(image, kvecs, klists=klists, return_gs=True, dtype=dtype)
    lock = np.stack([q['lockin'] for q in gs])
    lock_o = np.stack([q['lockin'] for q in parts['gs']])
    for pi, q in enumerate(gs):
        assert np.array_equal(q['kidx'], np.zeros(shape, np.int32))       # index work: the only candidate wins everywhere
        assert np.array_equal(q['w'][0], np.full(shape, kvecs[pi][0])) and np.array_equal(q['w'][1], np.full(shape, kvecs[pi][1]))
    err = _px_errors(u, u_o, 20)
    lrel = float(np.abs(lock - lock_o).max() / np.abs(lock_o).max())
    _record('config1_512_k1_%s' % np.dtype(dtype).name, dict(err, lockin_rel=lrel, u_max_px=float(np.abs(u_o).max())))
    if dtype is np.float64:
        assert err['max_px'] < TOL.C1_F64['u'] and lrel < 1e-11
        assert np.abs(u[C1_SL] - g['k1_u']).max() < TOL.C1_F64['u']
        assert np.abs(lock[C1_SL] - g['k1_lockin']).max() < 1e-11 * np.abs(lock_o).max()
    else:
        assert err['max_px'] < TOL.C1_F32['u_px'] and lrel < TOL.C1_F32['lockin_rel']
        d = u[C1_SL] - g['k1_u']
        assert np.abs(d - d.mean(axis=(1, 2), keepdims=True)).max() < TOL.C1_F32['u_px']
    # the reference test's own bar for this kind of field (tests/test_geometric_phase_analysis.py:61-66)
    assert np.all(np.abs(-u - u_true)[:, 20:-20, 20:-20] < 0.9)


# ---- configs[1] and configs[2]: measured fp32 accuracy ------------------------------------------------------
def test_config2_2048_accuracy():
    """configs[1]: 2048^2, 3 x 8, fp32: device f64 and f32 against the oracle.  `f32` is the library default = the
    reference's stopping test alone (10 + 10 iterations: the mode bench.py's `value` times); `f32_floor` the opt-in
    F32_EPS_FLOOR=4e-6.  Bounds: tests/tolerances.py (<= 2 x what was measured on MI355X); achieved errors recorded."""
    n = 2048
    kvecs = hex_kvecs(0.1, 7.0)
    img = hex_moire((n, n), kvecs, gaussian_bump_displacement((n, n)), noise=0.1, seed=5)
    kw, sigma, _ = orc.derive_params(kvecs)
    klists = np.stack(explicit_klists(kvecs, kw, 4, 2))
    cores = os.cpu_count() or 1
    u_ref, parts = orc.extract_displacement_field(img, kvecs, sigma=sigma, klists=klists, return_parts=True,
                                                  workers=cores, pool=min(cores, 8))
    ref_kidx = np.stack([g['kidx'] for g in parts['gs']])
    img0 = img - img.mean()
    rec = {'u_max_px': float(np.abs(u_ref).max())}
    out = {}
    for name, dtype, floor in (('f64', np.float64, None), ('f32', np.float32, None), ('f32_floor', np.float32, '4e-6')):
        _set_floor(floor)
        plan = _lib.Plan((n, n), 24, dtype)
        u, _, kidx, iters = plan.extract_displacement_field(img, kvecs, klists, sigma, 2 * sigma, want_kidx=True)
        plan.close()
        _set_floor(None)
        out[name] = u
        rec[name] = dict(_px_errors(u, u_ref, 2 * sigma), iters=list(iters))
        rec[name]['kidx_mismatch'] = _check_kidx_ties(kidx, ref_kidx, img0, klists, sigma,
                                                      TOL.F64['tie_rel'] if dtype is np.float64 else TOL.F32['tie_rel'])
    rec['f32_forced10'] = rec['f32']          # (the key of rounds 2-4: the same run since the floor became opt-in)
    _record('config2_2048_3x8_vs_oracle', rec)
    assert rec['f64']['kidx_mismatch'] == TOL.F64['kidx_frac']
    assert rec['f64']['max_px'] < TOL.F64['max_px']
    assert rec['f64']['iters'] == [10, 10]
    # the headline's mode: the reference's stopping test alone
    assert rec['f32']['iters'] == [10, 10]
    assert rec['f32']['kidx_mismatch'] <= TOL.F32['kidx_frac']
    assert rec['f32']['max_px'] < TOL.F32_C2['max_px'] and rec['f32']['rms_px'] < TOL.F32_C2['rms_px']
    # the opt-in early stop stays within the general f32 statement
    assert rec['f32_floor']['max_px'] < TOL.F32['max_px'] and rec['f32_floor']['rms_px'] < TOL.F32['rms_px']


@pytest.mark.parametrize('shape', [(1080, 1920), (1000, 1000), (1280, 1024)])
def test_non_power_of_two_frames_vs_oracle(shape):
    """camera frame sizes that are not powers of two (HD 1080 x 1920, the reference's decade sizes, 1280 x 1024):
    compact extension in the sweep, mixed-radix fused unwrap (transform-free columns when square) -- the whole
    driver in f64 against the oracle (kidx exact up to ties, u to 1e-6 px), f32 within the config-2 bounds"""
    kvecs = hex_kvecs(0.1, 7.0)
    img = hex_moire(shape, kvecs, gaussian_bump_displacement(shape), noise=0.1, seed=9)
    kw, sigma, _ = orc.derive_params(kvecs)
    klists = np.stack(explicit_klists(kvecs, kw, 3, 2))
    cores = os.cpu_count() or 1
    u_ref, parts = orc.extract_displacement_field(img, kvecs, sigma=sigma, klists=klists, return_parts=True,
                                                  workers=cores, pool=min(cores, 8))
    ref_kidx = np.stack([g['kidx'] for g in parts['gs']])
    img0 = img - img.mean()
    rec = {'u_max_px': float(np.abs(u_ref).max())}
    for name, dtype in (('f64', np.float64), ('f32', np.float32)):
        plan = _lib.Plan(shape, 18, dtype)
        u, _, kidx, iters = plan.extract_displacement_field(img, kvecs, klists, sigma, 2 * sigma, want_kidx=True)
        rec[name] = dict(_px_errors(u, u_ref, 2 * sigma), iters=list(iters), fft_len=[plan.fft_len(0), plan.fft_len(1)])
        rec[name]['kidx_mismatch'] = _check_kidx_ties(kidx, ref_kidx, img0, klists, sigma, 1e-12 if dtype is np.float64 else 4e-6)
        plan.close()
    _record('frame_%dx%d_3x6_vs_oracle' % shape, rec)
    assert rec['f64']['kidx_mismatch'] == 0.0
    assert rec['f64']['max_px'] < 1e-6
    assert rec['f32']['interior_max_px'] < 0.02 and rec['f32']['max_px'] < 0.1


def test_config3_4096_accuracy_f32_f64_oracle():
    """configs[2]: 4096^2, 3 x 16 + weighted unwrap -- the headline workload.  (1) device f64 against the ORACLE at full
    size; (2) fp32 in the headline's own mode (library default = the reference's stopping test, 10 + 10 iterations)
    against fp64 AND against the oracle, and with the opt-in residual floor (9 + 8); errors in pixels, recorded.
    |u| reaches ~155 px; the reference's own bar for this field is 0.9 px.  Bounds: tests/tolerances.py."""
    n = 4096
    kvecs = hex_kvecs(0.1, 7.0)
    img = hex_moire((n, n), kvecs, gaussian_bump_displacement((n, n)), noise=0.1, seed=100)
    kw, sigma, _ = orc.derive_params(kvecs)
    klists = np.stack(explicit_klists(kvecs, kw, 4, 4))
    out, rec = {}, {}
    for name, dtype, floor in (('f64', np.float64, None), ('f32', np.float32, None), ('f32_floor', np.float32, '4e-6')):
        _set_floor(floor)
        plan = _lib.Plan((n, n), 48, dtype)
        out[name] = plan.extract_displacement_field(img, kvecs, klists, sigma, 2 * sigma, want_lockins=True, want_kidx=True)
        plan.close()
        _set_floor(None)
    u64, l64, k64, it64 = out['f64']
    assert it64 == (10, 10)
    rec['u_max_px'] = float(np.abs(u64).max())
    for name in ('f32', 'f32_floor'):
        u32, l32, k32, it32 = out[name]
        rec[name + '_vs_f64'] = dict(_px_errors(u32, u64, 2 * sigma), iters=list(it32),
                                     kidx_mismatch=float((k32 != k64).mean()))
        same = k32 == k64
        rec[name + '_vs_f64']['lockin_rel'] = float(np.abs(l32 - l64)[same].max() / np.abs(l64).max())
    rec['f32_forced10_vs_f64'] = rec['f32_vs_f64']      # (the key of rounds 2-4)
    del l64
    # (1) the oracle on every host core (candidates through a thread pool)
    cores = os.cpu_count() or 1
    t = time.time()
    u_ref, parts = orc.extract_displacement_field(img, kvecs, sigma=sigma, klists=klists, return_parts=True,
                                                  workers=cores, pool=min(cores, 16))
    rec['oracle_seconds'] = time.time() - t
    ref_kidx = np.stack([g['kidx'] for g in parts['gs']])
    rec['f64_vs_oracle'] = dict(_px_errors(u64, u_ref, 2 * sigma))
    img0 = img - img.mean()
    rec['f64_vs_oracle']['kidx_mismatch'] = _check_kidx_ties(k64, ref_kidx, img0, klists, sigma, TOL.F64['tie_rel'])
    for name in ('f32', 'f32_floor'):
        rec[name + '_vs_oracle'] = dict(_px_errors(out[name][0], u_ref, 2 * sigma))
        rec[name + '_vs_oracle']['kidx_mismatch'] = _check_kidx_ties(out[name][2], ref_kidx, img0, klists, sigma, TOL.F32['tie_rel'])
    _record('config3_4096_3x16', rec)
    assert rec['f64_vs_oracle']['kidx_mismatch'] == TOL.F64['kidx_frac']
    assert rec['f64_vs_oracle']['max_px'] < TOL.F64['max_px']
    # the headline's mode, against f64 and against the oracle
    assert rec['f32_vs_f64']['iters'] == [10, 10]
    assert rec['f32_vs_f64']['lockin_rel'] < TOL.F32['lockin_rel']
    for key in ('f32_vs_f64', 'f32_vs_oracle'):
        assert rec[key]['kidx_mismatch'] <= TOL.F32['kidx_frac'], key
        assert rec[key]['max_px'] < TOL.F32_C3['max_px'] and rec[key]['rms_px'] < TOL.F32_C3['rms_px'], key
    for key in ('f32_floor_vs_f64', 'f32_floor_vs_oracle'):
        assert rec[key]['max_px'] < TOL.F32['max_px'] and rec[key]['rms_px'] < TOL.F32['rms_px'], key


# ---- configs[3]: 8192^2 in halo tiles --------------------------------------------------------------------
def test_config4_8192_tiled_vs_whole_image():
    """configs[3] on one GPU (all tiles on rank 0): 8192^2, 3 x 16, fp32, 2048^2 power-of-two windows with a
    3 sigma halo (25 tiles), global 8192^2 unwrap.  Tile interiors must agree with the whole-image run of
    the same device kernels within 0.05 px (the free mean of each component removed)."""
    from pygpa_amd import distributed as D
    n = 8192
    kvecs = hex_kvecs(0.1, 7.0)
    img = hex_moire((n, n), kvecs, gaussian_bump_displacement((n, n)), noise=0.05, seed=41, dtype=np.float32)
    klists = explicit_klists(kvecs, 0.04, 4, 4)
    t = time.time()
    u = D.extract_displacement_field_tiled(img, kvecs, sigma=10, klists=klists, halo=30, window=(2048, 2048), dtype=np.float32)
    t_first = time.time() - t
    t = time.time()
    u = D.extract_displacement_field_tiled(img, kvecs, sigma=10, klists=klists, halo=30, window=(2048, 2048), dtype=np.float32)
    t_tiled = time.time() - t
    assert np.isfinite(u).all()
    plan = _lib.Plan((n, n), 48, np.float32)
    uw = plan.extract_displacement_field(img, kvecs, np.stack(klists), 10, 20, 10)[0]
    t = time.time()
    uw = plan.extract_displacement_field(img, kvecs, np.stack(klists), 10, 20, 10)[0]
    t_whole = time.time() - t
    plan.close()
    err = _px_errors(u, uw, 64)
    _record('config4_8192_tiled_vs_whole', dict(err, seconds_tiled_incl_pcie=t_tiled, seconds_first_call=t_first,
                                                seconds_whole_incl_pcie=t_whole, tiles=25, window=2048, halo=30,
                                                u_max_px=float(np.abs(uw).max())))
    assert err['interior_max_px'] < 0.05


def test_tiled_reduced_grid_vs_oracle_same_tiling():
    """the config-4 tiling on a reduced grid (1024 x 1536, 512^2 windows, halo 30) against the oracle run on
    the SAME tiling, f64 and f32"""
    from pygpa_amd import distributed as D
    import test_distributed as TD
    shape = (1024, 1536)
    kvecs = hex_kvecs(0.1, 7.0)
    img = hex_moire(shape, kvecs, gaussian_bump_displacement(shape), noise=0.05, seed=17)
    klists = explicit_klists(kvecs, 0.04, 2, 2)
    u_ref = D.extract_displacement_field_tiled(img, kvecs, sigma=10, klists=klists, halo=30, window=(512, 512),
                                               compute=TD._oracle_compute())
    u = D.extract_displacement_field_tiled(img, kvecs, sigma=10, klists=klists, halo=30, window=(512, 512))
    assert np.abs(u - u_ref).max() < 1e-7
    u32 = D.extract_displacement_field_tiled(img, kvecs, sigma=10, klists=klists, halo=30, window=(512, 512), dtype=np.float32)
    e = _px_errors(u32, u_ref, 20)
    _record('tiled_1024x1536_f32_vs_oracle_same_tiling', e)
    assert e['max_px'] < 0.05


def test_tile_stage_stores_interiors_directly(gpa_option):
    """the tile stage's least-squares kernel stores the interior pixels straight into the tile blocks (default) instead of
    full-window fields + a copy of the interiors (NO_TILEFUSE=1): the same field bit for bit, ragged last tiles included
    (1000 x 1400 in 512^2 windows: interiors of 452, 96 and 44 pixels), f32 and f64"""
    from pygpa_amd import distributed as D
    shape = (1000, 1400)
    kvecs = hex_kvecs(0.1, 7.0)
    img = hex_moire(shape, kvecs, gaussian_bump_displacement(shape), noise=0.05, seed=23)
    klists = explicit_klists(kvecs, 0.04, 2, 2)
    for dtype in (np.float64, np.float32):
        u = D.extract_displacement_field_tiled(img, kvecs, sigma=10, klists=klists, halo=30, window=(512, 512), dtype=dtype)
        gpa_option('NO_TILEFUSE', '1')
        u_c = D.extract_displacement_field_tiled(img, kvecs, sigma=10, klists=klists, halo=30, window=(512, 512), dtype=dtype)
        gpa_option('NO_TILEFUSE', None)
        assert np.array_equal(u, u_c), float(np.abs(u - u_c).max())


# ---- configs[4]: 16384^2 tile grid + Lawler-Fujita, end to end --------------------------------------------
def test_config5_16384_tiled_undistort_end_to_end():
    """configs[4] on one GPU: 16384^2, 3 x 16, fp32, 81 windows of 2048^2, global 16384^2 unwrap, then
    undistort_image (Lawler-Fujita).  No oracle runs at this size; the checks are the reference's own
    end-to-end properties (tests/test_geometric_phase_analysis.py:61-78): -u recovers the true displacement
    within 0.9 px away from the border, and the undistorted image is the undeformed lattice within 2 %."""
    from pygpa_amd import distributed as D
    n = 16384
    kvecs = hex_kvecs(0.1, 7.0)
    # (built in row blocks: a 16384^2 float64 temporary is 2 GiB)
    img = np.empty((n, n), dtype=np.float32)
    orig = np.empty((n, n), dtype=np.float32)
    ux = np.empty((n, n), dtype=np.float32)
    y = (np.arange(n) - n // 2)[None, :].astype(np.float64)
    for r0 in range(0, n, 1024):
        x = (np.arange(r0, r0 + 1024) - n // 2)[:, None].astype(np.float64)
        # a bump of at most ~25 px so that the 35 fixed-point rounds of the inversion converge everywhere
        b = 0.05 * x * np.exp(-0.5 * ((x / (n / 8.0)) ** 2 + 1.2 * (y / (n / 6.0)) ** 2))
        ux[r0:r0 + 1024] = b
        acc, acc0 = np.zeros((1024, n)), np.zeros((1024, n))
        for kx, ky in kvecs:
            acc += np.cos(2 * np.pi * (kx * (x + b) + ky * y))
            acc0 += np.cos(2 * np.pi * (kx * x + ky * y))
        img[r0:r0 + 1024] = acc
        orig[r0:r0 + 1024] = acc0
    klists = explicit_klists(kvecs, 0.04, 4, 4)
    t = time.time()
    u = D.extract_displacement_field_tiled(img, kvecs, sigma=10, klists=klists, halo=30, window=(2048, 2048), dtype=np.float32)
    t_first = time.time() - t
    t = time.time()
    u = D.extract_displacement_field_tiled(img, kvecs, sigma=10, klists=klists, halo=30, window=(2048, 2048), dtype=np.float32)
    t_tiled = time.time() - t
    assert np.isfinite(u).all()
    b = 64
    d0 = -u[0] - ux
    d0 -= d0[b:-b, b:-b].mean()
    d1 = -u[1]
    d1 -= d1[b:-b, b:-b].mean()
    e0, e1 = float(np.abs(d0[b:-b, b:-b]).max()), float(np.abs(d1[b:-b, b:-b]).max())
    plan = _lib.Plan((n, n), 1, np.float32)
    t = time.time()
    rec_img = plan.undistort_image(img, -u)      # the extracted field is -u_true (tests/test_geometric_phase_analysis.py:63)
    t_lf = time.time() - t
    plan.close()
    dev = np.abs(rec_img - orig)[256:-256, 256:-256] / np.abs(orig).max()
    _record('config5_16384_tiled_undistort', {'seconds_tiled_incl_pcie': t_tiled, 'seconds_first_call': t_first,
                                              'seconds_undistort_incl_pcie': t_lf, 'tiles': 81, 'window': 2048,
                                              'u_err_max_px': [e0, e1], 'undistort_max_dev_frac': float(dev.max()),
                                              'u_true_max_px': float(np.abs(ux).max())})
    assert e0 < 0.9 and e1 < 0.9
    assert dev.max() < 0.02


# ---- N > 1 device path: two ranks sharing GPU 0 (gloo, host-staged collectives) ---------------------------
def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize('world', [1, 2, 3])
def test_pipelined_stream_ranks_one_gpu(tmp_path, world):
    """TiledPipeline.run_stream on the device with 1, 2 and 3 ranks sharing cuda:0 (gloo, host-staged gathers and
    hand-over): five images, the unwrap of image i on ranks (2 i + c) % N beside the sweeps of image i + 1; on its
    owner every field equals step() of the same image bit for bit, and each arrives exactly once"""
    for dtype in ('float64', 'float32'):
        port = _free_port()
        prefix = str(tmp_path / ('stream_%d_%s' % (world, dtype)))
        procs = []
        for r in range(world):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR='127.0.0.1',
                       MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
            procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', '_stream_rank_worker.py'), prefix, dtype],
                                          env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
        outs = [p.communicate(timeout=900)[0].decode() for p in procs]
        assert all(p.returncode == 0 for p in procs), '\n'.join(outs)
        res = [np.load(prefix + '_rank%d.npz' % r) for r in range(world)]
        assert all(bool(g['ok']) for g in res), (world, dtype)
        assert sorted(np.concatenate([g['seen'] for g in res]).tolist()) == [0, 1, 2, 3, 4]
        for g in res[1:]:
            assert np.array_equal(g['ref0'], res[0]['ref0'])      # step() itself: every rank holds the same field


def test_stream_resident_sources_order_tile_stage_after_stitch():
    """one rank, sources resident (run_stream([img] + [None] * k)): the stitch of image i reads the
    tile buffer IN PLACE on the component plans' streams while the tile stage of image i + 1 writes it on the tile plan's
    stream -- the tile plan now waits for the stitches (stitch_to_tiles).  A large stitch (4096^2, 9 windows) beside a
    short tile stage (2 x 1 candidates): every image of the stream must equal step() bit for bit."""
    from pygpa_amd import distributed as D
    n = 4096
    kvecs = hex_kvecs(0.1, 7.0)
    img = hex_moire((n, n), kvecs, gaussian_bump_displacement((n, n)), noise=0.05, seed=77, dtype=np.float32)
    klists = np.stack(explicit_klists(kvecs, 0.04, 2, 1))
    pipe = D.TiledPipeline((n, n), kvecs, klists, 10, 30, kmax=3, dtype=np.float32, window=(2048, 2048))
    pipe.load(img)
    ref = pipe.step().cpu().numpy().copy()
    got = []
    iters = pipe.run_stream([None] * 6, on_result=lambda i, u: got.append(u.cpu().numpy().copy()))
    pipe.close()
    assert len(got) == 6 and all(it == (3, 3) for it in iters)
    for i, u in enumerate(got):
        assert np.array_equal(u, ref), i


def test_tiled_two_ranks_one_gpu(tmp_path):
    """pygpa_amd.distributed.TiledPipeline with world size 2: windows dealt to two processes that share
    cuda:0, all_reduce / all_gather / broadcast staged through the host on gloo.  Every rank must hold the
    field of the single-process device run of the same tiling (the whole-image mean is reduced in another
    order: 1e-10), for both tilings and both precisions."""
    from pygpa_amd import distributed as D
    import test_distributed as TD
    img, kvecs, klists = TD._case()
    for dtype in ('float64', 'float32'):
        port = _free_port()
        prefix = str(tmp_path / ('tiled_' + dtype))
        procs = []
        for r in range(2):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE='2', LOCAL_RANK=str(r), MASTER_ADDR='127.0.0.1',
                       MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
            procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', '_tiled_rank_worker.py'), prefix, dtype],
                                          env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
        outs = [p.communicate(timeout=900)[0].decode() for p in procs]
        assert all(p.returncode == 0 for p in procs), '\n'.join(outs)
        u_w = D.extract_displacement_field_tiled(img, kvecs, klists=klists, halo=20, window=(64, 128), dtype=np.dtype(dtype))
        u_g = D.extract_displacement_field_tiled(img, kvecs, (2, 2), klists=klists, halo=20, dtype=np.dtype(dtype))
        tol = 1e-10 if dtype == 'float64' else 2e-4
        ntiles = 0
        for r in range(2):
            g = np.load(prefix + '_rank%d.npz' % r)
            assert np.abs(g['u'] - u_w).max() <= tol, (dtype, r)
            assert np.abs(g['u2'] - u_g).max() <= tol, (dtype, r)
            ntiles += int(g['tiles'])
        assert ntiles == 4
        g0, g1 = (np.load(prefix + '_rank%d.npz' % r) for r in range(2))
        assert np.array_equal(g0['u'], g1['u']) and np.array_equal(g0['u2'], g1['u2'])
        # the tile-sharded Lawler-Fujita stage (TiledPipeline.undistort: every rank the windows of its own tiles, one
        # all_reduce of the zero-filled outputs): both ranks hold what ONE whole-grid call makes of the same field, bit for bit
        plan = _lib.Plan(img.shape, 1, np.dtype(dtype))
        ref_lf = plan.undistort_image(img.astype(dtype), g0['u2'])
        ref_uinv = plan.invert_u_overlap(-g0['u2'])
        plan.close()
        for g in (g0, g1):
            assert np.array_equal(g['lf'], ref_lf, equal_nan=True), dtype
            assert np.array_equal(g['uinv'], ref_uinv, equal_nan=True), dtype


def test_tiled_nccl_single_rank(tmp_path):
    """the device-tensor collectives of the N > 1 path (all_reduce / all_gather_into_tensor / broadcast on RCCL)
    with the one rank a single-GPU box allows: same field as the torch-free single-process run"""
    from pygpa_amd import distributed as D
    import test_distributed as TD
    img, kvecs, klists = TD._case()
    prefix = str(tmp_path / 'nccl1')
    env = dict(os.environ, RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1',
               MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', '_tiled_rank_worker.py'), prefix, 'float64', 'nccl'],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert r.returncode == 0, r.stdout.decode()[-3000:]
    g = np.load(prefix + '_rank0.npz')
    u_w = D.extract_displacement_field_tiled(img, kvecs, klists=klists, halo=20, window=(64, 128), dtype=np.float64)
    u_g = D.extract_displacement_field_tiled(img, kvecs, (2, 2), klists=klists, halo=20, dtype=np.float64)
    assert np.abs(g['u'] - u_w).max() <= 1e-10 and np.abs(g['u2'] - u_g).max() <= 1e-10
    assert int(g['tiles']) == 4


def test_stream_and_step_on_nccl_single_rank_forced_collectives(tmp_path):
    """`run_stream` (and step()) on the nccl backend.  A single-GPU box allows ONE RCCL rank; with
    GPA_DIST_FORCE_COLLECTIVES=1 that rank takes every collective of the N > 1 schedule anyway -- all_reduce of the mean,
    all_gather_into_tensor of the tile blocks, the gathers to the owners, the broadcasts -- on device tensors, ordered against
    the library's streams through events (tiles_to_torch / torch_to_tiles): run_stream's fields equal step()'s bit for bit
    (the worker asserts it) and both equal the run without a process group; once more with GPA_DIST_SYNC=1 fences"""
    import test_distributed as TD
    from pygpa_amd import distributed as D
    _, kvecs, klists = TD._case()
    images = TD._stream_images(5)
    pipe = D.TiledPipeline(images[0].shape, kvecs, np.stack(klists), 6, 20, kmax=10, dtype=np.float32, device=0, grid=(2, 2))
    pipe.load(images[0])
    ref0 = pipe.step().cpu().numpy().copy()
    pipe.close()
    for sync in ('0', '1'):
        prefix = str(tmp_path / ('nccl_stream' + sync))
        env = dict(os.environ, RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()),
                   HSA_ENABLE_IPC_MODE_LEGACY='0', GPA_DIST_FORCE_COLLECTIVES='1', GPA_DIST_SYNC=sync)
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', '_stream_rank_worker.py'), prefix, 'float32', 'nccl'],
                           env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
        assert r.returncode == 0, r.stdout.decode()[-3000:]
        g = np.load(prefix + '_rank0.npz')
        assert bool(g['ok']) and list(g['seen']) == [0, 1, 2, 3, 4]
        assert np.array_equal(g['ref0'], ref0)


def test_stack_frames_over_two_ranks_one_gpu(tmp_path):
    """pygpa_amd.distributed.extract_displacement_field_stack_sharded with world size 2 (both ranks on cuda:0,
    gloo): blocks of frames per rank, no data-path collective, the gathered field equals the single-process
    stack call frame for frame, bit for bit"""
    from pygpa_amd import distributed as D
    import test_distributed as TD
    frames, kvecs, klists = TD._stack_case()
    for dtype in ('float64', 'float32'):
        port = _free_port()
        prefix = str(tmp_path / ('stack_' + dtype))
        procs = []
        for r in range(2):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE='2', LOCAL_RANK=str(r), MASTER_ADDR='127.0.0.1',
                       MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
            procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', '_stack_rank_worker.py'), prefix, dtype],
                                          env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
        outs = [p.communicate(timeout=900)[0].decode() for p in procs]
        assert all(p.returncode == 0 for p in procs), '\n'.join(outs)
        u1 = D.extract_displacement_field_stack_sharded(frames, kvecs, klists=klists, dtype=np.dtype(dtype))
        assert u1.shape == (5, 2) + frames.shape[1:]
        for r in range(2):
            assert np.array_equal(np.load(prefix + '_rank%d.npy' % r), u1), (dtype, r)


def test_bench_gpus_flag_spawns_ranks():
    """`python bench.py --gpus 2` (no launcher) must start 2 ranks itself and report n_gpus = 2 with the tile
    pipeline as the workload; here both ranks share GPU 0 over gloo and the image is small"""
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--size', '256',
           '--window', '128', '--kside', '2', '--backend', 'gloo', '--share-device']
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT')}
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['steps'] == 2 and out['value'] > 0
    assert out['config']['image'] == [512, 256] and out['config']['tiles'] >= 2
    assert out['config']['schedule'] == 'stream' and 'gather' in out['config']['collectives']
    assert set(out['config']['stage_ms_per_image_rank0']) == {'load', 'mean', 'tiles', 'gather', 'unwrap_wait', 'handover'}
    assert out['roofline'] is not None and out['roofline']['achieved'] > 0 and 'tile stage' in out['roofline']['note'], out.get('roofline_error')
    # the unpipelined schedule stays available
    r = subprocess.run(cmd + ['--schedule', 'step'], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    out = json.loads([ln for ln in r.stdout.decode().splitlines() if ln.startswith('{')][0])
    assert out['config']['schedule'] == 'step' and 'all_gather' in out['config']['collectives']
