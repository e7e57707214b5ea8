"""What the reference-shaped call costs from NumPy arrays: GPA.extract_displacement_field(image, ks) per image size."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pygpa_amd.geometric_phase_analysis as GPA
from pygpa_amd import _lib
from pygpa_amd.synthetic import explicit_klists, gaussian_bump_displacement, hex_kvecs, hex_moire
kvecs = hex_kvecs(0.1, 7.0)
kw = np.linalg.norm(kvecs, axis=1).mean() / 2.5
klists = explicit_klists(kvecs, kw, 4, 4)
for n in (512, 1024, 2048, 4096):
    img = hex_moire((n, n), kvecs, gaussian_bump_displacement((n, n)), noise=0.1, seed=1)
    for dt in (np.float32, np.float64):
        GPA.extract_displacement_field(img, kvecs, klists=klists, dtype=dt)
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            u = GPA.extract_displacement_field(img, kvecs, klists=klists, dtype=dt)
        dt_ms = (time.perf_counter() - t0) / reps * 1e3
        plan = _lib.get_plan((n, n), 48, dt)
        out = np.empty((2, n, n), dt)
        imgc = np.ascontiguousarray(img, dtype=dt)
        plan.extract_displacement_field(imgc, kvecs, np.stack(klists), 10, 20, out=out)
        t0 = time.perf_counter()
        for _ in range(reps):
            plan.extract_displacement_field(imgc, kvecs, np.stack(klists), 10, 20, out=out)
        dt2 = (time.perf_counter() - t0) / reps * 1e3
        print('%4d^2 %s  GPA.extract_displacement_field %.2f ms (%.0f Mpix/s)   plan call, converted input + out= reused: %.2f ms' %
              (n, np.dtype(dt).name, dt_ms, n * n / dt_ms / 1e3, dt2), flush=True)
