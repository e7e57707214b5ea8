import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from pygpa_amd import _lib
rng = np.random.default_rng(0)
shapes = [(300, 16384), (16384, 300), (384, 16384), (3000, 16384), (300, 8192), (300, 12000), (12000, 300), (300, 10000), (300, 9000), (300, 8200), (5000, 9000), (100, 16000), (100, 16380)]
for dt in (np.float32, np.float64):
    for shape in shapes:
        try:
            dx = rng.standard_normal((shape[0], shape[1] - 1)).astype(dt) * 0.1
            dy = rng.standard_normal((shape[0] - 1, shape[1])).astype(dt) * 0.1
            w = (0.5 + rng.random(shape)).astype(dt)
            plan = _lib.Plan(shape, 1, dt)
            phi, it = plan.unwrap_prediff(dx, dy, w, kmax=2)
            plan.close()
            print(np.dtype(dt).name, shape, 'ok', it, bool(np.isfinite(phi).all()), flush=True)
        except Exception as e:
            print(np.dtype(dt).name, shape, 'FAILS:', str(e)[:120], flush=True)
