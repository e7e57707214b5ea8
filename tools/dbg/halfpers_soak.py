"""soak of the persistent half-length row kernels: every shape REPS times through the persistent kernels (all runs equal bit for bit)
and once through the one-row-per-workgroup kernels (equal to them), kmax 10, with two solves of different plans in flight on the device"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'tests'))
from pygpa_amd import _lib
from test_gpu_unwrap_long import make_problem
REPS = int(os.environ.get('REPS', '12'))
bad = 0
t0 = time.time()
for shape in [(64, 8192), (128, 16384), (256, 8192), (2048, 8192), (1024, 16384), (4096, 8192), (8192, 8192), (64, 16384), (2048, 16384)]:
    dx, dy, w = make_problem(shape, seed=shape[0] + shape[1] + 1)
    dx, dy, w = (np.ascontiguousarray(v, dtype=np.float32) for v in (dx, dy, w))
    _lib.set_option('NO_ROWPERS', '1')
    plan = _lib.Plan(shape, 1, np.float32)
    ref, it = plan.unwrap_prediff(dx, dy, w, kmax=10)
    plan.close()
    _lib.set_option('NO_ROWPERS', None)
    plans = [_lib.Plan(shape, 1, np.float32) for _ in range(2)]
    n = 0
    for r in range(REPS):
        out, _ = plans[r & 1].unwrap_prediff(dx, dy, w, kmax=10)
        if not np.array_equal(out, ref):
            bad += 1
            n += 1
    for p in plans:
        p.close()
    print(shape, 'mismatches', n, 'of', REPS, flush=True)
print('total mismatches', bad, 'in %.0f s' % (time.time() - t0))
sys.exit(1 if bad else 0)
