"""Host-to-host throughput of Plan.extract_displacement_field_stack (NumPy frames in, u out): chunked pipeline
against one image per call of the reference-shaped driver.   python tools/stack_host_bench.py [--sizes 512,1024]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygpa_amd import _lib                                                     # noqa: E402
from pygpa_amd.synthetic import explicit_klists, gaussian_bump_displacement, hex_kvecs, hex_moire   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--sizes', default='512,1024')
ap.add_argument('--frames', type=int, default=128)
args = ap.parse_args()
kvecs = hex_kvecs(0.1, 7.0)
sigma = int(np.ceil(1 / np.linalg.norm(kvecs, axis=1).min()))
kw = np.linalg.norm(kvecs, axis=1).mean() / 2.5
klists = np.stack(explicit_klists(kvecs, kw, 4, 4))
for n in (int(v) for v in args.sizes.split(',')):
    img = hex_moire((n, n), kvecs, gaussian_bump_displacement((n, n)), noise=0.1, seed=1, dtype=np.float32)
    B = args.frames if n <= 512 else args.frames // 4
    frames = np.stack([img] * B)
    plan = _lib.Plan((n, n), 48, np.float32, device=0)
    plan.extract_displacement_field(img, kvecs, klists, sigma, 2 * sigma)
    t0 = time.perf_counter()
    for i in range(min(B, 16)):
        plan.extract_displacement_field(frames[i], kvecs, klists, sigma, 2 * sigma)
    t1 = (time.perf_counter() - t0) / min(B, 16)
    print('%d^2 one frame per call (host arrays):   %.3f ms/frame  %.0f Mpix/s' % (n, t1 * 1e3, n * n / t1 / 1e6), flush=True)
    for chunk in (None, B):
        plan.extract_displacement_field_stack(frames, kvecs, klists, sigma, 2 * sigma, chunk=chunk)   # (workspaces of this chunk size)
        t0 = time.perf_counter()
        plan.extract_displacement_field_stack(frames, kvecs, klists, sigma, 2 * sigma, chunk=chunk)
        dt = (time.perf_counter() - t0) / B
        print('%d^2 stack of %d, chunk=%s:   %.3f ms/frame  %.0f Mpix/s  (x%.2f)' % (n, B, chunk, dt * 1e3, n * n / dt / 1e6, t1 / dt), flush=True)
    plan.close()
