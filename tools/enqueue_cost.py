"""Host time to ENQUEUE one fused-driver call (gpa_extract_displacement_field_async returns before the GPU finishes)
against the time the GPU needs for it: is a small image bound by the host's launch rate?
usage: python tools/enqueue_cost.py [sizes...]"""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from pygpa_amd import _lib
from pygpa_amd.synthetic import hex_kvecs, hex_moire, explicit_klists, gaussian_bump_displacement

kvecs = hex_kvecs(0.1, 7.0)
klists = np.stack(explicit_klists(kvecs, 0.04, 4, 4))
for n in [int(a) for a in sys.argv[1:]] or [512, 1024, 2048, 4096]:
    img = hex_moire((n, n), kvecs, gaussian_bump_displacement((n, n)), noise=0.1, seed=1, dtype=np.float32)
    plan = _lib.Plan((n, n), 48, np.float32)
    d_img = _lib.DeviceBuffer(img.nbytes); d_img.upload(img)
    d_u = _lib.DeviceBuffer(2 * img.nbytes)
    for _ in range(3):
        plan.extract_displacement_field_async(d_img.ptr, kvecs, klists, 10, 20, 10, d_u.ptr)
    plan.sync()
    enq, tot = [], []
    for _ in range(30):
        t0 = time.perf_counter()
        plan.extract_displacement_field_async(d_img.ptr, kvecs, klists, 10, 20, 10, d_u.ptr)
        t1 = time.perf_counter()
        plan.sync()
        t2 = time.perf_counter()
        enq.append(t1 - t0); tot.append(t2 - t0)
    # back to back: the host runs ahead of the GPU as far as it can
    t0 = time.perf_counter()
    for _ in range(30):
        plan.extract_displacement_field_async(d_img.ptr, kvecs, klists, 10, 20, 10, d_u.ptr)
    t1 = time.perf_counter()
    plan.sync()
    t2 = time.perf_counter()
    print('%5d^2: enqueue %.3f ms (median), enqueue+wait %.3f ms; 30 calls back to back: host %.3f ms per call, all done after %.3f ms per call'
          % (n, 1e3 * np.median(enq), 1e3 * np.median(tot), 1e3 * (t1 - t0) / 30, 1e3 * (t2 - t0) / 30), flush=True)
    plan.close(); d_img.free(); d_u.free()
