import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'tests'))
from pygpa_amd import _lib
from test_gpu_unwrap_long import make_problem
for shape in [(64, 8192), (128, 8192), (192, 8192), (256, 8192), (512, 8192), (64, 16384), (128, 16384), (256, 16384)]:
    dx, dy, w = make_problem(shape, seed=5)
    dx, dy, w = (np.ascontiguousarray(v, dtype=np.float32) for v in (dx, dy, w))
    for kmax in (1, 3):
        out = {}
        for name, opt in (('pers', None), ('per-row', '1')):
            _lib.set_option('NO_ROWPERS', opt)
            plan = _lib.Plan(shape, 1, np.float32)
            plan.set_profiling(True)
            out[name], _ = plan.unwrap_prediff(dx, dy, w, kmax=kmax)
            prof = plan.last_kernel_profile()
            plan.close()
        _lib.set_option('NO_ROWPERS', None)
        d = np.abs(out['pers'] - out['per-row'])
        print(shape, 'kmax', kmax, 'max diff %.3e' % d.max(), 'n diff', int((d > 0).sum()), sorted(prof) if kmax == 1 else '', flush=True)
