#!/usr/bin/env python
"""One weighted unwrap component on resident data per image size and column-solve mode (COLSOLVE option): time per solve
(HIP events on the plan's stream), iteration count, per-kernel HIP-event times per working launch and the fraction of the
8 TB/s HBM peak each reaches on its algorithmic bytes.  python tools/unwrap_sizes.py --sizes 4096 8192 --modes default tri"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from pygpa_amd import _lib   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--sizes', type=int, nargs='+', default=[4096])
ap.add_argument('--modes', nargs='+', default=['default'])
ap.add_argument('--dtype', default='f32')
ap.add_argument('--kmax', type=int, default=10)
ap.add_argument('--chunk', default=None)
ap.add_argument('--reps', type=int, default=3)
a = ap.parse_args()
dt = np.float32 if a.dtype == 'f32' else np.float64
s = np.dtype(dt).itemsize
# algorithmic bytes per pixel and working launch
BPP = {'rowdct_fused_kernel': 3, 'colsolve_kernel': 2, 'colsolve_tri_kernel': 2, 'colstream_agg_kernel': 1, 'colstream_scan_kernel': 0,
       'colstream_apply_kernel': 2, 'rowidct_p_kernel': 3, 'pq_kernel': 3, 'rowidct_pq_kernel': 5, 'pqdct_kernel': 3}
_lib.set_option('F32_EPS_FLOOR', '0')
if a.chunk:
    _lib.set_option('COLSTREAM_CHUNK', a.chunk)
for n in a.sizes:
    rng = np.random.default_rng(n)
    x = np.arange(n, dtype=np.float32)[:, None] / n
    y = np.arange(n, dtype=np.float32)[None, :] / n
    psi = (40 * x + 25 * y + 6 * np.sin(6.28 * (1.5 * x + 0.5 * y))).astype(np.float32)
    psi += 0.05 * rng.standard_normal((n, n), dtype=np.float32)
    psi = (psi + np.pi) % (2 * np.pi) - np.pi
    dx = np.ascontiguousarray(np.diff(psi, axis=1), dtype=dt)
    dy = np.ascontiguousarray(np.diff(psi, axis=0), dtype=dt)
    w = (0.6 + 0.4 * np.cos(6.28 * 3 * x) * np.cos(6.28 * 2 * y)).astype(dt)
    del psi
    bufs = [_lib.DeviceBuffer(v.nbytes) for v in (dx, dy, w)]
    for b, v in zip(bufs, (dx, dy, w)):
        b.upload(v)
    phi = _lib.DeviceBuffer(n * n * s)
    for mode in a.modes:
        _lib.set_option('COLSOLVE', None if mode == 'default' else mode)
        plan = _lib.Plan((n, n), 1, dt)
        it = plan.unwrap_prediff_dev(bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, phi.ptr, kmax=a.kmax)
        plan.timer_start()
        for _ in range(a.reps):
            plan.unwrap_prediff_enqueue_dev(bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, phi.ptr, kmax=a.kmax)
        ms = plan.timer_stop() / a.reps
        plan.unwrap_finish()
        plan.set_profiling(True)
        it = plan.unwrap_prediff_dev(bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, phi.ptr, kmax=a.kmax)
        prof = plan.last_kernel_profile()
        plan.set_profiling(False)
        plan.close()
        line = []
        for k, (calls, tms) in prof.items():
            work = min(calls, it) if k in BPP else calls
            us = tms / max(work, 1) * 1e3
            frac = (' %.2f' % (BPP[k] * s * n * n / (us * 1e-6) / 8e12)) if BPP.get(k) else ''
            line.append('%s %.1f us%s' % (k.replace('_kernel', ''), us, frac))
        print('%6d^2 %s %-8s %8.3f ms/solve  it=%d  %.0f Mpix/s per component | %s' % (n, a.dtype, mode, ms, it, n * n / ms / 1e3, '; '.join(line)), flush=True)
    for b in bufs + [phi]:
        b.free()
_lib.set_option('COLSOLVE', None)
