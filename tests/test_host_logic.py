"""Host-side bookkeeping of the mirror modules, on the CPU: the control flow that sits above the C ABI is
run with the device call replaced by the oracle's array functions, and compared with the vectors the
real reference produced (tests/golden/peaks.npz, oracle/make_golden.py)."""
import numpy as np
import pytest

from oracle import gpa_oracle as orc

PEAK_CASES = ['clean128', 'noisy200x240', 'weak96', 'harmonics256', 'aniso160', 'stripe128']


class _OraclePeaks:
    """stands in for _lib.Plan.find_peaks: (coords, values) in skimage order, from the oracle"""
    calls = 0

    def find_peaks(self, image, sigma, dog_sigma, threshold_rel, want_smooth=False, max_out=4096):
        _OraclePeaks.calls += 1
        smooth = orc.smoothed_spectrum(image, sigma, DoG=dog_sigma > 0)
        c = orc.peak_local_max(smooth, threshold_rel)
        return c.astype(np.intp), smooth[tuple(c.T)]


@pytest.mark.parametrize('name', PEAK_CASES)
def test_extract_primary_ks_relaxation_loop(golden, name, monkeypatch):
    """the relaxation loop of the mirror returns what the reference's recursion returns, through every
    branch (plain, threshold / sigma relaxation -- 12 levels for the stripe image --, triangle selection)"""
    import pygpa_amd.geometric_phase_analysis as GPA
    monkeypatch.setattr(GPA, '_plan', lambda image, batch, dtype: _OraclePeaks())
    g = golden('peaks')
    thr, dog = g[name + '_kw']
    _OraclePeaks.calls = 0
    pks, aks = GPA.extract_primary_ks(g[name + '_image'], threshold=float(thr), DoG=bool(dog))
    assert np.array_equal(pks, g[name + '_primary'])
    assert np.array_equal(aks, g[name + '_all'])
    if name == 'stripe128':
        assert _OraclePeaks.calls > 5


def test_k_vector_helpers_match_oracle():
    """remove_negative_duplicates / select_closest_to_triangle / threshold schedule against the oracle's
    restatement (itself pinned by the peaks golden vectors) on random grid frequencies, including exact
    +k / -k pairs, near duplicates and closed triangles that tie to rounding"""
    import pygpa_amd.geometric_phase_analysis as GPA
    from pygpa_amd.mathtools import remove_negative_duplicates
    rng = np.random.default_rng(12)
    for _ in range(400):
        n = int(rng.integers(0, 12))
        ks = rng.integers(-20, 21, size=(n, 2)) / rng.choice([64., 100., 37.])
        if n:
            extra = -ks[rng.integers(0, n, size=3)] + rng.normal(size=(3, 2)) * 1e-7
            ks = np.concatenate([ks, extra])
        assert np.array_equal(remove_negative_duplicates(ks), orc.remove_negative_duplicates(ks))
        if len(ks) >= 3:
            assert np.array_equal(GPA.select_closest_to_triangle(ks), orc.select_closest_to_triangle(ks))
            assert np.array_equal(GPA.smallest_sum(ks[:3]), orc.smallest_sum(ks[:3]))
    assert np.isnan(GPA.smallest_sum(np.zeros((2, 2))))
    for t in list(rng.random(200)) + [0.7, 0.2, 0.1999999, 0.001, 0.0005]:
        nxt = GPA._lower_threshold(t)
        assert (nxt is None and orc._decrease_threshold(t) == t) or nxt == orc._decrease_threshold(t)


def test_generate_klists_matches_reference(golden):
    """generate_klists (host bookkeeping for wfr3 / wfr4) against the list the reference produced"""
    import pygpa_amd.geometric_phase_analysis as GPA
    g = golden('variants_64')
    lists = GPA.generate_klists(g['kvecs'], kmax=1.12, kmin=0.9, sort_list=True)
    assert len(lists) == 3 and np.array_equal(lists[1], g['wfr4_ring'])
    # unsorted: the same points in grid order
    raw = GPA.generate_klists(g['kvecs'], kmax=1.12, kmin=0.9)[1]
    assert sorted(map(tuple, raw)) == sorted(map(tuple, g['wfr4_ring']))
