"""whole-driver time and per-kernel profile for rectangular camera frames (device-resident, f32, 3 x 16 candidates)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygpa_amd import _lib
from pygpa_amd.synthetic import explicit_klists, gaussian_bump_displacement, hex_kvecs, hex_moire
kvecs = hex_kvecs(0.1, 7.0)
kw = np.linalg.norm(kvecs, axis=1).mean() / 2.5
klists = np.stack(explicit_klists(kvecs, kw, 4, 4))
for shape in ((1080, 1920), (1280, 1024), (1392, 1040), (2160, 2560), (1024, 1024), (2048, 2048)):
    img = hex_moire(shape, kvecs, gaussian_bump_displacement(shape), noise=0.1, seed=1, dtype=np.float32)
    plan = _lib.Plan(shape, 48, np.float32)
    d = _lib.DeviceBuffer(img.nbytes); u = _lib.DeviceBuffer(2 * img.nbytes)
    d.upload(img)
    for _ in range(3):
        plan.extract_displacement_field_async(d.ptr, kvecs, klists, 10, 20, 10, u.ptr)
    plan.sync()
    t0 = time.perf_counter()
    for _ in range(20):
        plan.extract_displacement_field_async(d.ptr, kvecs, klists, 10, 20, 10, u.ptr)
    plan.sync()
    dt = (time.perf_counter() - t0) / 20
    plan.set_profiling(True)
    plan.extract_displacement_field_dev(d.ptr, kvecs, klists, 10, 20, 10, u.ptr)
    prof = plan.last_kernel_profile()
    print(shape, 'fft', plan.fft_len(0), plan.fft_len(1), '%.3f ms  %.0f Mpix/s |' % (dt * 1e3, shape[0] * shape[1] / dt / 1e6),
          ' '.join('%s %.0f' % (k.replace('_kernel', ''), ms / c * 1e3) for k, (c, ms) in prof.items() if c), flush=True)
    plan.close(); d.free(); u.free()
