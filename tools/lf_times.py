#!/usr/bin/env python
"""Lawler-Fujita undistortion on resident data (gpa_undistort_image_dev): kernel-only time per image size by HIP events on
the plan's stream, and the per-kernel table of a profiled call with each kernel's compulsory bytes.
    python tools/lf_times.py --sizes 4096 16384
Compulsory HBM bytes per pixel (s = bytes per real; padded grid m = n + 24 per side for mode 'nearest'):
  pad_edge        2 comps x (read s + write s)
  fir_rows/cols   per pass read s + write s; 2 comps x 2 passes for u, 1 x 2 for the image
  invert          36 rounds of 2 x 16 coefficient gathers per pixel: compulsory = the two coefficient fields once + u_inv out
                  = 2 s + 2 s (the gathers themselves are served by L1 / L2: 36 x 32 x s bytes per pixel of CACHE traffic)
  warp            coefficient field once + u_inv in + image out = s + 2 s + s"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from pygpa_amd import _lib   # noqa: E402
from pygpa_amd.synthetic import gaussian_bump_displacement, hex_kvecs, hex_moire   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--sizes', type=int, nargs='+', default=[4096])
ap.add_argument('--dtype', default='f32')
ap.add_argument('--reps', type=int, default=3)
ap.add_argument('--scale', type=float, default=None, help='displacement amplitude factor (default: n / 500, the bench field)')
a = ap.parse_args()
dt = np.float32 if a.dtype == 'f32' else np.float64
s = np.dtype(dt).itemsize
BYTES = {'pad_edge_kernel': 2, 'fir_rows_kernel': 2, 'fir_cols_kernel': 2, 'invert_kernel': 4, 'warp_constant_kernel': 4}
for n in a.sizes:
    shape = (n, n)
    ks = hex_kvecs(0.1, 7.0)
    u = (gaussian_bump_displacement(shape) * (a.scale if a.scale is not None else 1.0)).astype(dt)
    img = hex_moire(shape, ks, u.astype(np.float64) if n <= 4096 else None, dtype=dt)
    plan = _lib.Plan(shape, 1, dt)
    bufs = [_lib.DeviceBuffer(v.nbytes) for v in (u, img)]
    bufs[0].upload(u)
    bufs[1].upload(img)
    out = _lib.DeviceBuffer(n * n * s)
    uinv = _lib.DeviceBuffer(2 * n * n * s)
    plan.undistort_image_dev(bufs[1].ptr, bufs[0].ptr, out.ptr, uinv_ptr=uinv.ptr)
    plan.sync()
    plan.timer_start()
    for _ in range(a.reps):
        plan.undistort_image_dev(bufs[1].ptr, bufs[0].ptr, out.ptr, uinv_ptr=uinv.ptr)
    ms = plan.timer_stop() / a.reps
    plan.set_profiling(True)
    plan.undistort_image_dev(bufs[1].ptr, bufs[0].ptr, out.ptr, uinv_ptr=uinv.ptr)
    prof = plan.last_kernel_profile()
    plan.set_profiling(False)
    line = []
    for k, (calls, tms) in prof.items():
        gb = BYTES.get(k, 0) * s * n * n * calls / 1e9
        line.append('%s x%d %.3f ms (%.0f GB/s = %.2f of 8 TB/s on %.2f GB)' % (k.replace('_kernel', ''), calls, tms, gb / (tms * 1e-3) if tms else 0,
                                                                               gb / (tms * 1e-3) / 8000 if tms else 0, gb))
    umax = float(np.abs(u).max())
    print('%6d^2 %s undistort_image_dev %.3f ms = %.0f Mpix/s  max|u| %.0f px | %s' % (n, a.dtype, ms, n * n / ms / 1e3, umax, '; '.join(line)), flush=True)
    plan.close()
    for b in bufs + [out, uinv]:
        b.free()
