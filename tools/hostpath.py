import sys, time
sys.path.insert(0, '.')
import numpy as np
from pygpa_amd import _lib
from pygpa_amd.synthetic import hex_kvecs, hex_moire, explicit_klists, gaussian_bump_displacement
n = 4096
kvecs = hex_kvecs(0.1, 7.0)
img = hex_moire((n, n), kvecs, gaussian_bump_displacement((n, n)), noise=0.1, seed=1, dtype=np.float32)
klists = np.stack(explicit_klists(kvecs, 0.04, 4, 4))
plan = _lib.Plan((n, n), 48, np.float32)
for rep in range(4):
    t = time.perf_counter()
    u = plan.extract_displacement_field(img, kvecs, klists, 10, 20, 10)[0]
    print('host-pointer extract_displacement_field: %.2f ms' % ((time.perf_counter() - t) * 1e3))
import pygpa_amd.geometric_phase_analysis as GPA
for rep in range(3):
    t = time.perf_counter()
    u = GPA.extract_displacement_field(img, kvecs, klists=list(klists), dtype=np.float32)
    print('mirror GPA.extract_displacement_field: %.2f ms' % ((time.perf_counter() - t) * 1e3))

pimg = _lib.pinned_empty((n, n), np.float32)
pimg[...] = img
pu = _lib.pinned_empty((2, n, n), np.float32)
for rep in range(4):
    t = time.perf_counter()
    plan.extract_displacement_field(pimg, kvecs, klists, 10, 20, 10, out=pu)
    print('pinned in / pinned out: %.2f ms' % ((time.perf_counter() - t) * 1e3))
assert np.array_equal(pu, u) or np.abs(pu - u).max() < 1e-3
del pimg, pu
