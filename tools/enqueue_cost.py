"""host enqueue time vs GPU time of the fused driver at small sizes"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from pygpa_amd import _lib
from pygpa_amd.synthetic import hex_kvecs, hex_moire, explicit_klists, gaussian_bump_displacement
for n in (512, 1024, 2048, 4096):
    kvecs = hex_kvecs(0.1, 7.0)
    img = hex_moire((n, n), kvecs, gaussian_bump_displacement((n, n)), noise=0.1, seed=1, dtype=np.float32)
    klists = np.stack(explicit_klists(kvecs, 0.04, 4, 4))
    plan = _lib.Plan((n, n), 48, np.float32)
    d_img = _lib.DeviceBuffer(img.nbytes); d_img.upload(img)
    d_u = _lib.DeviceBuffer(2 * img.nbytes)
    for _ in range(3):
        plan.extract_displacement_field_async(d_img.ptr, kvecs, klists, 10, 20, 10, d_u.ptr)
    plan.sync()
    K = 20
    t0 = time.perf_counter()
    for _ in range(K):
        plan.extract_displacement_field_async(d_img.ptr, kvecs, klists, 10, 20, 10, d_u.ptr)
    t1 = time.perf_counter()
    plan.sync()
    t2 = time.perf_counter()
    print('%5d^2: host enqueue %.3f ms/call, total %.3f ms/call' % (n, (t1 - t0) / K * 1e3, (t2 - t0) / K * 1e3), flush=True)
    plan.close()
