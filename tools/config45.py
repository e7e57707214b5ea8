"""BASELINE configs 4-5 at full size on ONE GPU (all tiles dealt to rank 0): timing + sanity.
usage: python tools/config45.py SIZE G0xG1 [--lf] [--whole]"""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from pygpa_amd import distributed as D, _lib
from pygpa_amd.synthetic import hex_kvecs, hex_moire, explicit_klists, gaussian_bump_displacement

n = int(sys.argv[1])
g = None if sys.argv[2] == 'auto' else tuple(int(x) for x in sys.argv[2].split('x'))
window = None
for a in sys.argv[3:]:
    if a.startswith('--window='):
        window = int(a.split('=')[1])
kvecs = hex_kvecs(0.1, 7.0)
t = time.time()
u_true = gaussian_bump_displacement((n, n)).astype(np.float32)
img = hex_moire((n, n), kvecs, u_true, dtype=np.float32)
print('image %d^2 generated in %.1f s' % (n, time.time() - t), flush=True)
klists = explicit_klists(kvecs, 0.1 / 2.5, 4, 4)
kw = dict(window=(window, window)) if window else {}
if '--torch' in sys.argv:
    import torch
    kw['_force_torch'] = True
if '--host' in sys.argv:
    kw['compute'] = D.default_compute((window, window) if window else (n // g[0] + 64, n // g[1] + 64), (n, n), 48, np.float32, 0)
us = {}
for rep in range(3):
    t = time.time()
    u = D.extract_displacement_field_tiled(img, kvecs, g, sigma=10, klists=klists, halo=32, kmax=10, dtype=np.float32, **kw)
    dt = time.time() - t
    print('tiled %s grid %s window %s: %.2f s  (%.1f Mpix/s incl. host staging)' % (img.shape, g, window, dt, n * n / dt / 1e6), flush=True)
print('u finite', np.isfinite(u).all(), 'rms', float(np.sqrt((u ** 2).mean())))
if '--whole' in sys.argv:
    # whole-image extraction on the same GPU: tile interiors must agree up to the free mean of each component
    p = _lib.Plan((n, n), 48, np.float32)
    for rep in range(2):
        t = time.time()
        uw = p.extract_displacement_field(img - img.mean(), kvecs, klists, 10, 20, 10)[0]
        print('whole image %d^2: %.2f s' % (n, time.time() - t), flush=True)
    p.close()
    d = (u - u.mean(axis=(1, 2), keepdims=True)) - (uw - uw.mean(axis=(1, 2), keepdims=True))
    print('tiled - whole: max |d| = %.3e px, rms %.3e px (|u| max %.1f px)' % (np.abs(d).max(), np.sqrt((d ** 2).mean()), np.abs(uw).max()))
if '--lf' in sys.argv:
    p = _lib.Plan((n, n), 1, np.float32)
    for rep in range(2):
        t = time.time()
        out = p.undistort_image(img, u)
        print('undistort_image %d^2: %.2f s' % (n, time.time() - t), flush=True)
    print('undistorted finite', np.isfinite(out).all())
