#!/usr/bin/env python
"""where the gap between `value` (D2H of u inside the step) and `resident_only` comes from: the step loop of bench.py with
the download cut to a given number of bytes (0 = none, full = 2 n^2 s)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
from pygpa_amd import _lib
from pygpa_amd.synthetic import hex_kvecs, explicit_klists
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
kvecs = hex_kvecs(0.1, 7.0)
sigma = int(np.ceil(1 / np.linalg.norm(kvecs, axis=1).min()))
kw = np.linalg.norm(kvecs, axis=1).mean() / 2.5
klists = np.stack(explicit_klists(kvecs, kw, 4, 4))
_lib.set_option('F32_EPS_FLOOR', '0')
g = bench.SingleGPU(n, 3, 16, np.float32, kvecs, klists, sigma, 10)
full = 2 * n * n * 4
for nbytes in (0, 4096, full // 16, full // 2, full):
    def run(steps):
        for i in range(steps):
            j = i & 1
            if nbytes:
                g.plan.download_wait(j)
            g.enqueue(0, j)
            if nbytes:
                g.plan.download_async(g.h_u[0][j].reshape(-1).view(np.uint8)[:nbytes], g.d_u[0][j].ptr, j)
        g.sync()
    run(3)
    t0 = time.perf_counter(); run(20); dt = time.perf_counter() - t0
    print('download %10d bytes per step: %.4f ms per step  %.1f Mpix/s' % (nbytes, dt / 20 * 1e3, n * n * 20 / dt / 1e6), flush=True)
g.close()
