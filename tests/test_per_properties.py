"""a9 (smooth + periodic DFT, moisan2011.per -- parity unpinned, the third-party source is absent): an independent check of
the oracle's `per` that shares no code with it (tests/per_properties.py: Moisan's defining equations in the spatial
domain).  The GPU twin of this test is tests/test_gpu_hypothesis.py::test_per_dft_defining_properties."""
import numpy as np
import pytest

from oracle import gpa_oracle as orc
from per_properties import check_decomposition


@pytest.mark.parametrize('shape', [(37, 52), (64, 64), (63, 65), (100, 30)])
def test_oracle_per_satisfies_moisans_equations(shape):
    rng = np.random.default_rng(3)
    ramp = np.linspace(0, 3, shape[1])[None, :] + np.linspace(-1, 2, shape[0])[:, None] ** 2
    u = rng.standard_normal(shape) + ramp
    phat = orc.per(u, inverse_dft=False)[0]
    check_decomposition(u, phat, 1e-12, rng)
    # an image whose opposite border rows / columns are EQUAL (period N - 1, not N) has no smooth component
    x, y = np.meshgrid(np.arange(shape[0]), np.arange(shape[1]), indexing='ij')
    per_img = np.cos(2 * np.pi * 3 * x / (shape[0] - 1)) + np.sin(2 * np.pi * 2 * y / (shape[1] - 1))
    assert np.abs(orc.per(per_img, inverse_dft=False)[0] - np.fft.fft2(per_img)).max() < 1e-9
