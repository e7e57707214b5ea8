import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from oracle import gpa_oracle as orc
from pygpa_amd import _lib
from pygpa_amd.synthetic import hex_kvecs, hex_moire, explicit_klists, gaussian_bump_displacement
shape, r_k, xi, nx, ny, sigma = (257, 320), 0.175, 52.0, 1, 2, 8
import test_gpu_parity as T
for c in T._random_cases(12345, 150):
    if c[0] == shape and abs(c[1] - r_k) < 1e-3:
        shape, r_k, xi, nx, ny, sigma, seed = c
        break
kvecs = hex_kvecs(r_k, xi)
img = hex_moire(shape, kvecs, 0.4 * gaussian_bump_displacement(shape), noise=0.2, seed=seed % 1000)
kw = np.linalg.norm(kvecs, axis=1).mean() / 2.5
klists = np.stack(explicit_klists(kvecs, kw, nx, ny))
u_ref, parts = orc.extract_displacement_field(img, kvecs, sigma=sigma, klists=klists, return_parts=True)
print('oracle iters', parts['iters'])
p64 = _lib.Plan(shape, 3 * nx * ny, np.float64)
u64, _, _, it64 = p64.extract_displacement_field(img, kvecs, klists, sigma, 2 * sigma)
p32 = _lib.Plan(shape, 3 * nx * ny, np.float32)
u32, _, _, it32 = p32.extract_displacement_field(img, kvecs, klists, sigma, 2 * sigma)
rel = lambda a, b: np.abs(a - b).max() / np.abs(b).max()
print('floor', os.environ.get('GPA_F32_EPS_FLOOR'), 'it64', it64, 'it32', it32, 'rel64', rel(u64, u_ref), 'rel32', rel(u32, u_ref))
d = np.abs(u32 - u_ref)
dr = 2 * sigma
print('max |d| whole %.3e px, interior %.3e px, rms whole %.3e, |u|max %.1f' % (d.max(), d[:, dr:-dr, dr:-dr].max(), np.sqrt((d**2).mean()), np.abs(u_ref).max()))
w = parts['weights']
print('where is the max:', np.unravel_index(d.argmax(), d.shape), 'shape', d.shape)
