"""The reference's hypothesis-driven tests restated with hypothesis on the drop-in modules (GPU), and the independent
check of the a9 kernel:
  tests/test_phase_unwrap.py:11-31, :49-75   (kmax drawn from 1..30, N = 256 ramp)
  tests/test_geometric_phase_analysis.py:44-58 (extract_primary_ks on lattices drawn over r_k, theta, psi, kappa)
The lattice generator is this build's (latticegen is absent): an isotropic hexagonal lattice strained by kappa along the
direction psi; what the reference asserts -- every extracted k within 1.5 / size of a generating k -- does not depend
on latticegen's conventions."""
import numpy as np
import pytest
from hypothesis import example, given, settings, HealthCheck
import hypothesis.strategies as st

import pygpa_amd.geometric_phase_analysis as GPA
import pygpa_amd.phase_unwrap as pu
from pygpa_amd import _lib
from pygpa_amd.synthetic import hex_kvecs, hex_moire
from per_properties import check_decomposition

pytestmark = pytest.mark.gpu
# derandomize: the same draws on every run (a GPU suite that fails one run in ten teaches nothing)
COMMON = dict(deadline=None, derandomize=True, suppress_health_check=[HealthCheck.too_slow, HealthCheck.data_too_large])


@settings(max_examples=20, **COMMON)
@given(kmax=st.integers(1, 30))
def test_equivalent_phase_unwrap_ref_phase_unwrap(kmax):
    """reference tests/test_phase_unwrap.py:11-31"""
    N = 256
    xx, yy = np.meshgrid(np.arange(N), np.arange(N), indexing='ij')
    psi0 = (yy + xx) / (4 * np.sqrt(2))
    psi = pu._wrapToPi(psi0)
    weight = np.ones_like(psi)
    res_ref = pu.phase_unwrap_ref(psi=psi, weight=weight, kmax=kmax)
    assert np.allclose(res_ref - res_ref.mean(), psi0 - psi0.mean())
    assert np.allclose(res_ref, pu.phase_unwrap(psi=psi, weight=weight, kmax=kmax))
    assert np.allclose(res_ref, pu.phase_unwrap(psi=psi, weight=None, kmax=kmax))


@settings(max_examples=20, **COMMON)
@given(kmax=st.integers(1, 30))
def test_equivalent_phase_unwrap_ref_prediff_phase_unwrap_prediff(kmax):
    """reference tests/test_phase_unwrap.py:49-75"""
    N = 256
    xx, yy = np.meshgrid(np.arange(N), np.arange(N), indexing='ij')
    psi0 = (yy + xx) / (4 * np.sqrt(2))
    psi = pu._wrapToPi(psi0)
    dx = np.diff(psi, axis=1)
    dy = np.diff(psi, axis=0)
    weight = np.ones_like(psi)
    res_ref = pu.phase_unwrap_ref_prediff(dx=dx, dy=dy, weight=weight, kmax=kmax)
    assert np.allclose(res_ref - res_ref.mean(), psi0 - psi0.mean())
    assert np.allclose(res_ref, pu.phase_unwrap_prediff(dx=dx, dy=dy, weight=weight, kmax=kmax))
    assert np.allclose(res_ref, pu.phase_unwrap_prediff(dx=dx, dy=dy, weight=None, kmax=kmax))
    assert np.allclose(pu.phase_unwrap_ref(psi=psi, weight=weight, kmax=kmax), res_ref)


def strained_ks(r_k, theta, psi, kappa):
    """six k-vectors of a hexagonal lattice of pitch 1 / r_k rotated by theta, compressed by 1 / kappa along the
    direction psi (degrees)"""
    ks = hex_kvecs(r_k, theta, n=6)
    c, s = np.cos(np.deg2rad(psi)), np.sin(np.deg2rad(psi))
    rot = np.array([[c, s], [-s, c]])
    return ks @ (rot.T @ np.diag([1.0 / kappa, 1.0]) @ rot)


@settings(max_examples=40, **COMMON)
@given(theta=st.floats(0., 60.), psi=st.floats(-90., 90.), kappa=st.floats(1. + 1e-7, 2, exclude_min=True),
       r_k=st.floats(0.03, 0.24))
@example(r_k=0.03, theta=1.625, psi=1.75, kappa=1.65625)   # found by hypothesis: the reference's algorithm itself misses here
def test_extract_primary_ks(r_k, theta, psi, kappa):
    """reference tests/test_geometric_phase_analysis.py:44-58, as a PARITY property: on every drawn lattice the device
    finds exactly the k-vectors the oracle's restatement of the reference driver finds (grid frequencies: equal), and
    wherever those satisfy the reference's accuracy bar -- every extracted k within 1.5 / size of a generating k -- so
    do the device's.  (The bar itself is a property of the reference's peak finder, not of this build: with this
    build's lattice generator it fails at the corner of the parameter box, r_k = 0.03 strained by kappa = 1.66, where
    the smallest k is 2.3 bins from DC -- for the oracle exactly as for the device.)"""
    from oracle import gpa_oracle as orc
    size = 128
    ori_ks = strained_ks(r_k, theta, psi, kappa)
    original = hex_moire((size, size), ori_ks[:3])
    ext_ks, _ = GPA.extract_primary_ks(original, DoG=False)
    ref_ks, _ = orc.extract_primary_ks(original, DoG=False)
    assert ext_ks.shape == ref_ks.shape and np.abs(ext_ks - ref_ks).max() < 1e-12
    abs_diffs = np.linalg.norm((ext_ks[None] - ori_ks[:, None]), axis=-1).min(axis=0)
    ref_diffs = np.linalg.norm((ref_ks[None] - ori_ks[:, None]), axis=-1).min(axis=0)
    if np.all(ref_diffs < 1.5 / size):
        assert np.all(abs_diffs < 1.5 / size)
    if r_k >= 0.06:      # away from that corner the bar holds outright
        assert np.all(abs_diffs < 1.5 / size)


@pytest.mark.parametrize('dtype,tol', [(np.float64, 1e-11), (np.float32, 3e-6)])
@pytest.mark.parametrize('shape', [(64, 64), (63, 65), (100, 30), (200, 300)])
def test_per_dft_defining_properties(shape, dtype, tol):
    """gpa_per_dft against Moisan's defining equations in the spatial domain (tests/per_properties.py): no code shared
    with oracle/gpa_oracle.py:per, whose formula the kernels follow (a9 stays 'parity unpinned' against moisan2011
    itself, which this image does not have)"""
    rng = np.random.default_rng(8)
    ramp = np.linspace(0, 3, shape[1])[None, :] + np.linspace(-1, 2, shape[0])[:, None] ** 2
    u = (rng.standard_normal(shape) + ramp).astype(dtype)
    plan = _lib.Plan(shape, 1, dtype)
    phat = plan.per_dft(u)
    # the components themselves from the device (gpa_per, inverse_dft=True) through the same checks: their DFT is taken
    # here, on the host, only to feed the checker
    pc, sc = plan.per(u, inverse_dft=True)
    plan.close()
    check_decomposition(u.astype(np.float64), phat, tol, rng)
    assert np.abs(pc.astype(np.float64) + sc - u).max() < 4 * tol * np.abs(u).max()
    check_decomposition(u.astype(np.float64), np.fft.fft2(pc.astype(np.float64)), 4 * tol, rng)


@pytest.mark.parametrize('dtype,tol', [(np.float64, 1e-10), (np.float32, 3e-4)])
def test_invert_u_modes_vs_scipy(dtype, tol):
    """`mode=` of invert_u / invert_u_overlap (geometric_phase_analysis.py:248, :262): 'nearest' and 'constant' against the
    same scipy.ndimage.map_coordinates calls the reference makes (oracle), including the cval = nan of the last round of
    the overlap variant; any other scipy mode is refused"""
    from oracle import gpa_oracle as orc
    rng = np.random.default_rng(5)
    shape = (72, 90)
    x, y = np.meshgrid(np.arange(shape[0]) - 36.0, np.arange(shape[1]) - 45.0, indexing='ij')
    us = np.stack([2.5 * np.exp(-(x ** 2 + y ** 2) / 400.0) + 0.02 * x, 1.5 * np.sin(y / 14.0) + 0.3]) + 0.01 * rng.standard_normal((2,) + shape)
    for mode in ('nearest', 'constant'):
        # (an ODD iteration count with edge = 0 leaves border pixels of invert_u out of range in the last round:
        #  they must come back as 0 -- only invert_u_overlap ends on a cval = nan round)
        for kw in (dict(iters=6, edge=0), dict(iters=4, edge=3), dict(iters=5, edge=0), dict(iters=35, edge=0)):
            for fn, ofn in ((GPA.invert_u_overlap, orc.invert_u_overlap), (GPA.invert_u, orc.invert_u)):
                ref = ofn(us, mode=mode, **kw)
                out = fn(us, mode=mode, dtype=dtype, **kw)
                assert out.shape == ref.shape
                assert np.array_equal(np.isnan(out), np.isnan(ref)), (mode, kw, fn.__name__)
                ok = ~np.isnan(ref)
                assert np.abs(out[ok] - ref[ok]).max() < tol * max(1.0, np.abs(ref[ok]).max()), (mode, kw, fn.__name__)
    with pytest.raises(NotImplementedError):
        GPA.invert_u(us, mode='wrap', dtype=dtype)


def test_stack_ragged_chunks_and_fallback(monkeypatch, gpa_option):
    """the batched unwrap workspace is a capacity (a ragged last chunk and a shorter second stack reuse it);
    a shape without a batched unwrap (here: the mixed-radix engine switched off on a 100 x 60 frame) falls back to
    per-frame calls instead of raising -- both give the per-frame loop's numbers"""
    gpa_option('NO_LAT', '1')
    kvecs = hex_kvecs(0.12, 11.0)
    shape = (128, 96)
    frames = np.stack([hex_moire(shape, kvecs, None, noise=0.1, seed=s) for s in range(5)])
    kw = np.linalg.norm(kvecs, axis=1).mean() / 2.5
    from pygpa_amd.synthetic import explicit_klists
    klists = np.stack(explicit_klists(kvecs, kw, 2, 2))
    plan = _lib.Plan(shape, 12, np.float64)
    ref = np.stack([plan.extract_displacement_field(f, kvecs, klists, 8, 16)[0] for f in frames])
    u, it = plan.extract_displacement_field_stack(frames, kvecs, klists, 8, 16, chunk=2)      # chunks of 2, 2, 1
    assert np.array_equal(u, ref) and it.shape == (5, 2)
    u3, _ = plan.extract_displacement_field_stack(frames[:3], kvecs, klists, 8, 16, chunk=3)  # grows the capacity
    u1, _ = plan.extract_displacement_field_stack(frames[3:4], kvecs, klists, 8, 16)           # and a shorter stack reuses it
    assert np.array_equal(u3, ref[:3]) and np.array_equal(u1, ref[3:4])
    plan.close()
    gpa_option('NO_MR', '1')
    shape = (100, 60)
    frames = np.stack([hex_moire(shape, kvecs, None, noise=0.1, seed=s) for s in range(3)])
    plan = _lib.Plan(shape, 12, np.float64)
    ref = np.stack([plan.extract_displacement_field(f, kvecs, klists, 8, 16)[0] for f in frames])
    u, it = plan.extract_displacement_field_stack(frames, kvecs, klists, 8, 16)
    assert np.array_equal(u, ref) and np.all(it > 0)
    plan.close()
