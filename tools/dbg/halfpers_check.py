import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'tests'))
from pygpa_amd import _lib
from oracle import gpa_oracle as orc
from test_gpu_unwrap_long import make_problem
def rel(a, b): return float(np.abs(a - b).max() / np.abs(b).max())
def rms(a, b): return float(np.sqrt(np.mean((a - b) ** 2)) / np.abs(b).max())
shape = (512, 16384)
for seed in (1, 2, 3, 4):
    dx, dy, w = make_problem(shape, seed=seed)
    dx, dy, w = (np.ascontiguousarray(v, dtype=np.float32) for v in (dx, dy, w))
    for kmax in (1, 10):
        ref = orc.unwrap_prediff(dx.astype(np.float64), dy.astype(np.float64), w.astype(np.float64), kmax=kmax, compat=False)
        out = {}
        for name, opt in (('pers', None), ('per-row', '1')):
            _lib.set_option('NO_ROWPERS', opt)
            plan = _lib.Plan(shape, 1, np.float32)
            out[name], _ = plan.unwrap_prediff(dx, dy, w, kmax=kmax)
            plan.close()
        _lib.set_option('NO_ROWPERS', None)
        d = np.abs(out['pers'] - out['per-row'])
        i = np.unravel_index(np.argmax(d), d.shape)
        print(seed, kmax, 'vs oracle: pers max %.2e rms %.2e | per-row max %.2e rms %.2e | pers vs per-row max %.2e at %s, rowmax of diff: %s' % (
            rel(out['pers'], ref), rms(out['pers'], ref), rel(out['per-row'], ref), rms(out['per-row'], ref), rel(out['pers'], out['per-row']), i,
            np.round(d.max(axis=1)[::64] * 1e5, 1)), flush=True)
