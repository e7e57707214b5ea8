"""pass A / pass B on a padded axis against a periodic one at the same transform length and the same number of rows:
(4096, 3900) [y padded, L = 4096], (3900, 4096) [x padded], (4096, 4096)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygpa_amd import _lib
from pygpa_amd.synthetic import explicit_klists, gaussian_bump_displacement, hex_kvecs, hex_moire
kvecs = hex_kvecs(0.1, 7.0)
kw = np.linalg.norm(kvecs, axis=1).mean() / 2.5
klists = np.stack(explicit_klists(kvecs, kw, 4, 4))
for shape in ((4096, 4096), (4096, 3900), (3900, 4096), (3900, 3900)):
    img = hex_moire(shape, kvecs, gaussian_bump_displacement(shape), noise=0.1, seed=1, dtype=np.float32)
    plan = _lib.Plan(shape, 48, np.float32)
    d = _lib.DeviceBuffer(img.nbytes); u = _lib.DeviceBuffer(2 * img.nbytes)
    d.upload(img)
    plan.extract_displacement_field_dev(d.ptr, kvecs, klists, 10, 20, 10, u.ptr)
    plan.set_profiling(True)
    acc = {}
    for _ in range(3):
        plan.extract_displacement_field_dev(d.ptr, kvecs, klists, 10, 20, 10, u.ptr)
        for name, (calls, ms) in plan.last_kernel_profile().items():
            acc[name] = acc.get(name, 0.0) + ms / 3
    print(shape, 'fft', plan.fft_len(0), plan.fft_len(1), ' '.join('%s %.3f' % (k.replace('_kernel', ''), v) for k, v in acc.items() if k in ('passA_kernel', 'passB_kernel')), flush=True)
    plan.close(); d.free(); u.free()
