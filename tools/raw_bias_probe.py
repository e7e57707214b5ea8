#!/usr/bin/env python
"""diagnosis: bias of the gradient stage with raw winners against the compensating run (f32), per component"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from pygpa_amd import _lib
from pygpa_amd.synthetic import hex_kvecs, gaussian_bump_displacement, hex_moire, explicit_klists
from oracle import gpa_oracle as orc
shape = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 2048)
knx, kny = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (4, 2)
kvecs = hex_kvecs(0.1, 7.0)
img = hex_moire(shape, kvecs, gaussian_bump_displacement(shape), noise=0.1, seed=5)
kw, sigma, _ = orc.derive_params(kvecs)
klists = np.stack(explicit_klists(kvecs, kw, knx, kny))
if len(sys.argv) > 5:
    sigma = float(sys.argv[5])
res = {}
for dt in (np.float64, np.float32):
    for raw in (True, False):
        _lib.set_option('NO_RAW', None if raw else '1')
        plan = _lib.Plan(shape, 3 * knx * kny, dt)
        res[(dt, raw)] = plan.extract_gradients(img.astype(dt), kvecs, klists, sigma, 2 * sigma)
        plan.close()
_lib.set_option('NO_RAW', None)
b = 2 * int(sigma) + 4
ref = res[(np.float64, False)]
for raw in (True, False):
    g = res[(np.float32, raw)]
    for name, a, r in (('dudx', g[0], ref[0]), ('dudy', g[1], ref[1])):
        d = (a.astype(np.float64) - r)[:, b:-b, b:-b]
        print('f32 raw=%d %s  mean %s  rms %s' % (raw, name, d.mean(axis=(1, 2)), np.sqrt((d ** 2).mean(axis=(1, 2)))))
d64 = [np.abs(a - r).max() for a, r in zip(res[(np.float64, True)], ref)]
print('f64 raw vs compensated max', d64)
