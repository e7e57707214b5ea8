#!/usr/bin/env python
"""The rows either side of the hot path, one call each on a synthetic image: host wall clock of the
NumPy-in / NumPy-out entry point (copies included) -- run it under `rocprofv3 --kernel-trace --stats` (tools/gpu_kstats.sh
next<N>) for the per-kernel device times.
    python tools/next_rows.py --sizes 4096 --dtype f32"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from pygpa_amd import _lib   # noqa: E402
from pygpa_amd.synthetic import gaussian_bump_displacement, hex_kvecs, hex_moire   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--sizes', type=int, nargs='+', default=[4096])
ap.add_argument('--dtype', default='f32')
ap.add_argument('--what', nargs='+', default=['per', 'peaks', 'deconv', 'jac', 'plane'])
a = ap.parse_args()
dt = np.float32 if a.dtype == 'f32' else np.float64


def clock(f, reps=2):
    f()
    t = time.perf_counter()
    for _ in range(reps):
        r = f()
    return (time.perf_counter() - t) / reps * 1e3, r


for n in a.sizes:
    shape = (n, n)
    ks = hex_kvecs(0.1, 7.0)
    u = gaussian_bump_displacement(shape) if n <= 4096 else None
    img = hex_moire(shape, ks, u, dtype=dt)
    img = img - img.mean()
    plan = _lib.Plan(shape, 3, dt)
    out = []
    if 'per' in a.what:
        ms, _ = clock(lambda: plan.per_dft(img))
        out.append('per_dft %.1f ms' % ms)
    if 'peaks' in a.what:
        ms, r = clock(lambda: plan.find_peaks(img, 1.0, 50.0, 0.7))
        out.append('find_peaks(DoG) %.1f ms (%d candidates)' % (ms, len(r[0])))
        ms, r = clock(lambda: plan.find_peaks(img, 1.0, 0.0, 0.7))
        out.append('find_peaks(no DoG) %.1f ms' % ms)
    if 'jac' in a.what:
        rng = np.random.default_rng(0)
        grads = rng.standard_normal((3,) + shape + (2,)).astype(dt) * 0.1
        wts = rng.random((3,) + shape).astype(dt)
        ms, J = clock(lambda: plan.phasegradient2J(ks, grads, wts, 1.0))
        out.append('phasegradient2J %.1f ms' % ms)
        from pygpa_amd import property_extract as pe   # noqa: E402
        ms, _ = clock(lambda: pe.props_from_Jac(J))
        out.append('props_from_Jac %.1f ms' % ms)
        del grads, wts, J
    if 'plane' in a.what:
        ms, r = clock(lambda: plan.fit_plane(img))
        out.append('fit_plane %.1f ms (%d passes)' % (ms, r[1]))
    plan.close()
    if 'deconv' in a.what:
        dr, sigma = 20, 10.0
        pplan = _lib.Plan((n + 4 * dr, n + 4 * dr), 1, dt)
        field = (u[0] if u is not None else img).astype(dt)
        ms, _ = clock(lambda: pplan.gaussian_deconvolve(field, dr, sigma, 5000.0))
        out.append('gaussian_deconvolve %.1f ms' % ms)
        pplan.close()
    print('%6d^2 %s | %s' % (n, a.dtype, ' | '.join(out)), flush=True)
