import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'tests'))
from pygpa_amd import _lib
from oracle import gpa_oracle as orc
from test_gpu_unwrap_long import make_problem
def rel(a, b): return float(np.abs(a - b).max() / np.abs(b).max())
for shape in [(64, 16384), (128, 8192)]:
    ea, eb = [], []
    for seed in range(1, 13):
        dx, dy, w = make_problem(shape, seed=seed)
        ref = orc.unwrap_prediff(dx, dy, w, kmax=10, compat=False)
        out = {}
        for name, opt in (('pers', None), ('per-row', '1')):
            _lib.set_option('NO_ROWPERS', opt)
            plan = _lib.Plan(shape, 1, np.float32)
            out[name], _ = plan.unwrap_prediff(dx, dy, w, kmax=10)
            plan.close()
        _lib.set_option('NO_ROWPERS', None)
        ea.append(rel(out['pers'], ref)); eb.append(rel(out['per-row'], ref))
    print(shape, 'pers    :', ' '.join('%.1e' % e for e in ea), '| median %.2e max %.2e' % (np.median(ea), max(ea)))
    print(shape, 'per-row :', ' '.join('%.1e' % e for e in eb), '| median %.2e max %.2e' % (np.median(eb), max(eb)), flush=True)
