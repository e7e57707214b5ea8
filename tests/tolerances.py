"""The stated tolerances of the displacement-field driver (README.md "Tolerances" quotes this file; the full-size
GPU tests of tests/test_gpu_configs.py assert exactly these numbers).

Reference: extract_displacement_field, geometric_phase_analysis.py:907-932, computes in complex128 / float64.
Errors are in PIXELS of displacement, after removing the free mean of each component (phase_unwrap.py:110-114),
on fields with |u| up to ~155 px (BASELINE configs[2]) -- i.e. max 0.02 px is 1.3e-4 relative.

f64 build vs the oracle (the reference's arithmetic):   index work bit-exact, u to rounding.
f32 build (BASELINE configs[1-2] name fp32) vs the f64 build and vs the oracle, with the reference's stopping test
(the library default; 10 + 10 PCG iterations at configs[2]):
    rms  |du| <= 1e-5 px      (measured 3.3e-6 at 4096^2, 2.7e-6 at 2048^2)
    max  |du| <= 0.02 px      (measured 0.009 at 4096^2 and 0.0044 at 2048^2: ONE winner flip at an amplitude tie each)
    kidx may differ from the f64 / oracle winner at <= 2.5e-7 of the pixels (measured: ONE pixel of 16.8 M at 4096^2 =
    6e-8, one of 4.2 M at 2048^2 = 2.4e-7), and only where the two candidates' amplitudes tie to <= 4e-6 of the largest
    amplitude (conftest.kidx_mismatch_is_tie).
Per-configuration bounds below are <= 2 x what was measured on MI355X (profiles/r02_accuracy.json, re-measured by
every run of the tests into gpurun_out/r02_accuracy.json; round 5's copy: profiles/r05_accuracy.json)."""

F64 = dict(max_px=2e-11, kidx_frac=0.0, tie_rel=1e-12)        # measured 6.8e-12 px (4096^2), 4.4e-12 (2048^2)
F32 = dict(rms_px=1e-5, max_px=0.02, kidx_frac=2.5e-7, tie_rel=4e-6, lockin_rel=1.6e-6)   # lockin_rel measured 7.9e-7
F32_C2 = dict(rms_px=5.5e-6, max_px=0.0089)                   # configs[1], 2048^2 3 x 8: measured 2.74e-6 / 0.00443
F32_C3 = dict(rms_px=6.6e-6, max_px=0.0181)                   # configs[2], 4096^2 3 x 16: measured 3.28e-6 / 0.00904
# configs[0], 512^2, one reference k-vector per peak (iterate_GPA route and a K = 1 driver call), against the
# reference's own outputs (tests/golden/config1_512.npz); |u| up to 19 px, phases up to ~20 rad
C1_F64 = dict(corr=1e-12, prs=1e-8, u=1e-9)
C1_F32 = dict(corr=2e-7, prs_rel=2e-5, u_px=2e-3, lockin_rel=2e-6)
# configs[3]'s image (8192^2, 3 x 16, |u| up to 310 px) through the WHOLE-IMAGE driver against the oracle at full size
# (round 6, tests/test_gpu_pins.py): measured rms 8.7e-6 px, max 0.0041 px, 2 winner flips in 67 M pixels (3e-8)
F32_C4 = dict(rms_px=1.8e-5, max_px=0.0083, kidx_frac=2.5e-7)
# Lawler-Fujita at the benchmark's displacement (4096^2, |u| up to 155 px) against the oracle: u_inv in px, the undistorted
# image relative to its maximum; measured f64 2.7e-13 / 2.4e-13, f32 1.2e-4 / 8.8e-5
LF_4096 = {'float64': dict(u_inv_px=1e-9, rec_rel=1e-9), 'float32': dict(u_inv_px=2.5e-4, rec_rel=1.8e-4)}
# ... and at 16384^2 with |u| up to 621 px in f32 against the equation that defines the inverse: sample coordinates up to
# 16384 carry an f32 ulp of 9.8e-4 px; measured residual 7.6e-4 px, reconstruction within 0.014 of a lattice of amplitude 3
LF_16384_F32 = dict(residual_px=2e-3, rec_abs=0.03)
