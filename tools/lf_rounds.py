import os, sys, numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from pygpa_amd import _lib
from pygpa_amd.synthetic import gaussian_bump_displacement
for n in (4096,):
    u = gaussian_bump_displacement((n, n)).astype(np.float32)
    plan = _lib.Plan((n, n), 1, np.float32)
    du = _lib.DeviceBuffer(u.nbytes); du.upload(u)
    out = _lib.DeviceBuffer(u.nbytes)
    plan.invert_u_dev(du.ptr, out.ptr)
    plan.sync()
    r = np.empty((2, n, n), np.float32); out.download_into(r)
    rounds = r[0].astype(int)
    print('rounds histogram (pixels):', np.bincount(rounds.ravel(), minlength=37))
    print('mean rounds', rounds.mean(), ' pixels not fixed at exit', float((r[1] == 0).mean()))
    # fraction of 64-pixel wave rows (16x4 of a tile) whose exit was at the cap
    print('fraction at cap', float((rounds >= 35).mean()))
