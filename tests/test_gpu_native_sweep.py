"""The lock-in sweep on axes in NATIVE mode (pygpa_amd/csrc/gpa_sweep_mr.hip, option NATIVE=1): a smooth length that is not
a power of two is transformed at its own length on the mixed-radix engine instead of on a padded power of two.  The
reference's filter is the circular one of exactly that length (geometric_phase_analysis.py:72-75), so the native kernels
are held

  * to the oracle (the reference's own arithmetic) at the tolerances of tests/test_gpu_parity.py, kidx bit-exact up to
    amplitude ties, and
  * to the padded power-of-two path of the same library (the default) on every entry point that runs a sweep: all
    lock-ins, best-of-K, the gated selection of wfr4, selection + phase gradient, the fused driver, a stack of frames.

The mode is OPT-IN: it measured slower than the padded path at every size (profiles/r04_native_sweep_rejected.txt); these
tests keep the rejected kernels honest.  `gpa_plan_axis_native` says which axes a plan runs natively."""
import os

import numpy as np
import pytest

from conftest import kidx_mismatch_is_tie
from oracle import gpa_oracle as orc
from pygpa_amd import _lib
from pygpa_amd.synthetic import hex_kvecs, gaussian_bump_displacement, hex_moire, explicit_klists

pytestmark = pytest.mark.gpu

TOL = {
    np.float64: dict(lock=1e-11, grad=1e-10, pcg=1e-9, tie=1e-12),
    np.float32: dict(lock=2e-6, grad=2e-5, pcg=2e-5, tie=2e-6),
}
DTYPES = [np.float64, np.float32]


def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def check_kidx(kidx, ref_kidx, img0, klist, sigma, tie_tol):
    bad = kidx != ref_kidx
    if not bad.any():
        return
    amps = np.abs(orc.lockin_batch(img0, klist, sigma))
    assert np.all(kidx_mismatch_is_tie(amps, kidx, ref_kidx, tie_tol)[bad]), \
        '%d kidx mismatches that are not amplitude ties' % int(bad.sum())


def case(shape, seed=5, noise=0.2):
    kvecs = hex_kvecs(0.11, 4.0)
    img = hex_moire(shape, kvecs, gaussian_bump_displacement(shape), noise=noise, seed=seed)
    return kvecs, img - img.mean()


# (n0, n1, natively run axes at sigma 6): 500 = 4 5^3 (T 128, 4 samples per thread), 1000 = 10^3 (8), 96 / 60 short rows
# (one wavefront), 250 = 2 5^3, 154 = 14 11 and 130 = 10 13 (the generic odd-prime butterflies), 1500: padded 2048 < 1.5 n ->
# stays padded, 512: a power of two
SHAPES = [((500, 500), (True, True)), ((250, 1000), (True, True)), ((1000, 96), (True, True)), ((500, 512), (True, False)),
          ((60, 1500), (True, False)), ((154, 130), (True, True))]


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('shape,native', SHAPES)
def test_native_sweep_against_oracle_and_padded_path(shape, native, dtype, gpa_option):
    gpa_option('NATIVE', '1')
    kvecs, img0 = case(shape)
    sigma = 6
    klist = explicit_klists(kvecs, 0.03, 3, 3)[0]
    ref = orc.sweep(img0, sigma, klist, kvecs[0], workers=8)
    plan = _lib.Plan(shape, len(klist), dtype)
    lock, kidx, _ = plan.sweep(img0, kvecs[0], klist, sigma)
    assert (plan.axis_native(0), plan.axis_native(1)) == native
    check_kidx(kidx, ref['kidx'], img0, klist, sigma, TOL[dtype]['tie'])
    same = kidx == ref['kidx']
    assert same.mean() > 0.999
    assert rel(lock[same], ref['lockin'][same]) < TOL[dtype]['lock']
    # all lock-ins (no selection) through the same tables
    lb = plan.lockin_batch(img0, klist[:3], sigma)
    assert rel(lb, orc.lockin_batch(img0, klist[:3], sigma)) < TOL[dtype]['lock']
    # the padded power-of-two path of the same plan (tables restaged), then native again
    gpa_option('NATIVE', None)
    lock_p, kidx_p, _ = plan.sweep(img0, kvecs[0], klist, sigma)
    assert (plan.axis_native(0), plan.axis_native(1)) == (False, False)
    same = kidx == kidx_p
    assert same.mean() > 0.999 and rel(lock[same], lock_p[same]) < TOL[dtype]['lock']
    gpa_option('NATIVE', '1')
    lock2, kidx2, _ = plan.sweep(img0, kvecs[0], klist, sigma)
    assert (plan.axis_native(0), plan.axis_native(1)) == native
    assert np.array_equal(kidx2, kidx) and np.array_equal(lock2, lock)
    # another sigma on the same plan: filter tables of the native axes replaced
    ref9 = orc.sweep(img0, 9, klist, kvecs[0], workers=8)
    lock9, kidx9, _ = plan.sweep(img0, kvecs[0], klist, 9)
    check_kidx(kidx9, ref9['kidx'], img0, klist, 9, TOL[dtype]['tie'])
    same = kidx9 == ref9['kidx']
    assert rel(lock9[same], ref9['lockin'][same]) < TOL[dtype]['lock']
    plan.close()


@pytest.mark.parametrize('dtype', DTYPES)
def test_native_gated_and_gradient_modes(dtype, gpa_option):
    """the less travelled selection modes on native axes: wfr4's gated chain and selection + phase gradient
    (geometric_phase_analysis.py:839-862, :763-813) against the oracle and against the padded path"""
    gpa_option('NATIVE', '1')
    shape = (250, 500)
    kvecs, img0 = case(shape, seed=11)
    sigma = 7
    klist = explicit_klists(kvecs, 0.03, 5, 5)[0]
    K = len(klist)
    dk = np.linalg.norm(klist[1] - klist[0])
    gate = (np.linalg.norm(klist[:, None, :] - klist[None, :, :], axis=-1) < 2 * np.sqrt(2) * dk + 1e-12)
    plan = _lib.Plan(shape, K, dtype)
    lg, kg = plan.sweep_gated(img0, kvecs[0], klist, sigma, gate)
    assert plan.axis_native(0) and plan.axis_native(1)
    for mode in (0, 1, 2):
        lock, kidx, grad = plan.sweep(img0, kvecs[0], klist, sigma, want_grad=True, grad_mode=mode)
        gpa_option('NATIVE', None)
        lock_p, kidx_p, grad_p = plan.sweep(img0, kvecs[0], klist, sigma, want_grad=True, grad_mode=mode)
        gpa_option('NATIVE', '1')
        same = kidx == kidx_p
        assert same.mean() > 0.999
        assert rel(lock[same], lock_p[same]) < TOL[dtype]['lock']
        ok = same & np.isfinite(grad[..., 0]) & np.isfinite(grad_p[..., 0]) & np.isfinite(grad[..., 1]) & np.isfinite(grad_p[..., 1])
        # (a pixel whose neighbour picked another candidate differs between the paths only where kidx does: compare where
        #  the 3 x 3 neighbourhood agrees)
        nb = same.copy()
        nb[1:, :] &= same[:-1, :]
        nb[:-1, :] &= same[1:, :]
        nb[:, 1:] &= same[:, :-1]
        nb[:, :-1] &= same[:, 1:]
        ok &= nb
        d = np.abs(grad[ok] - grad_p[ok])
        d = np.minimum(d, np.abs(d - np.pi))      # wrapToPi(2 g) / 2: values at +- pi / 2 are the same angle
        assert d.max() < (1e-9 if dtype is np.float64 else 2e-3)
    gpa_option('NATIVE', None)
    lg_p, kg_p = plan.sweep_gated(img0, kvecs[0], klist, sigma, gate)
    gpa_option('NATIVE', '1')
    same = kg == kg_p
    assert same.mean() > (0.999 if dtype is np.float64 else 0.97)     # f32 amplitude near-ties flip a gated chain
    assert rel(lg[same], lg_p[same]) < TOL[dtype]['lock']
    if dtype is np.float64:
        ref = orc.wfr4(img0, sigma, klist, kvecs[0], dk)
        same = kg == ref['kidx']
        assert same.mean() > 0.999 and rel(lg[same], ref['lockin'][same]) < TOL[dtype]['lock']
    plan.close()


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('shape', [(500, 500), (400, 500)])
def test_native_fused_driver_against_oracle(shape, dtype, gpa_option):
    """extract_displacement_field (geometric_phase_analysis.py:907-932) on native axes: small K takes the candidate-split
    pass B (PB_PART + merge) at 500^2, the plain per-candidate kernel otherwise; u against the oracle and the padded path"""
    gpa_option('NATIVE', '1')
    kvecs = hex_kvecs(0.1, 7.0)
    img = hex_moire(shape, kvecs, 0.6 * gaussian_bump_displacement(shape), noise=0.05, seed=3, dtype=dtype)
    klists = np.stack(explicit_klists(kvecs, 0.04, 2, 2))
    plan = _lib.Plan(shape, 12, dtype)
    u, _, _, iters = plan.extract_displacement_field(img, kvecs, klists, 10, 20, kmax=10)
    assert plan.axis_native(0) and plan.axis_native(1)
    cores = os.cpu_count() or 1
    ref_u = orc.extract_displacement_field(img.astype(np.float64), kvecs, sigma=10, klists=klists, workers=cores)
    bound = 1e-6 if dtype is np.float64 else 0.1          # px: the bounds of tests/test_gpu_configs.py
    assert np.abs(u - ref_u).max() < bound, float(np.abs(u - ref_u).max())
    gpa_option('NATIVE', None)
    u_p, _, _, iters_p = plan.extract_displacement_field(img, kvecs, klists, 10, 20, kmax=10)
    gpa_option('NATIVE', '1')
    assert np.abs(u - u_p).max() < bound
    # 16 candidates per peak (the per-row loop without the candidate split)
    klists16 = np.stack(explicit_klists(kvecs, 0.04, 4, 4))
    plan16 = _lib.Plan(shape, 48, dtype)
    u16, _, _, _ = plan16.extract_displacement_field(img, kvecs, klists16, 10, 20, kmax=10)
    ref16_u = orc.extract_displacement_field(img.astype(np.float64), kvecs, sigma=10, klists=klists16, workers=cores)
    assert np.abs(u16 - ref16_u).max() < bound, float(np.abs(u16 - ref16_u).max())
    plan.close()
    plan16.close()


def test_native_stack_of_frames_equals_single_images(gpa_option):
    """a stack of 500^2 frames (blockIdx.z / .y = image in the native kernels) equals the single-image driver"""
    gpa_option('NATIVE', '1')
    n = 500
    kvecs = hex_kvecs(0.1, 7.0)
    klists = np.stack(explicit_klists(kvecs, 0.04, 2, 2))
    imgs = np.stack([hex_moire((n, n), kvecs, (0.4 + 0.3 * i) * gaussian_bump_displacement((n, n)), noise=0.05, seed=i, dtype=np.float32)
                     for i in range(3)])
    plan = _lib.Plan((n, n), 12, np.float32)
    u_b, it_b = plan.extract_displacement_field_stack(imgs, kvecs, klists, 10, 20, kmax=10, chunk=3)
    assert plan.axis_native(0) and plan.axis_native(1)
    for i in range(3):
        u, _, _, iters = plan.extract_displacement_field(imgs[i], kvecs, klists, 10, 20, kmax=10)
        assert np.abs(u_b[i] - u).max() < 2e-4 * max(1.0, np.abs(u).max()), (i, float(np.abs(u_b[i] - u).max()))
    plan.close()
