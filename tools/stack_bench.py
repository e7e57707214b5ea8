"""Throughput of gpa_extract_displacement_field_batch_dev on stacks of small images (images and u resident in HBM).
    python tools/stack_bench.py [--sizes 256,512,1024] [--stacks 1,2,4,8,16,32]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygpa_amd import _lib                                                     # noqa: E402
from pygpa_amd.synthetic import explicit_klists, gaussian_bump_displacement, hex_kvecs, hex_moire   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--sizes', default='256,512,1024')
    ap.add_argument('--stacks', default='1,2,4,8,16,32')
    ap.add_argument('--reps', type=int, default=10)
    args = ap.parse_args()
    kvecs = hex_kvecs(0.1, 7.0)
    sigma = int(np.ceil(1 / np.linalg.norm(kvecs, axis=1).min()))
    kw = np.linalg.norm(kvecs, axis=1).mean() / 2.5
    klists = np.stack(explicit_klists(kvecs, kw, 4, 4))
    for n in (int(v) for v in args.sizes.split(',')):
        img = hex_moire((n, n), kvecs, gaussian_bump_displacement((n, n)), noise=0.1, seed=1, dtype=np.float32)
        plan = _lib.Plan((n, n), 48, np.float32, device=0)
        d1, u1 = _lib.DeviceBuffer(img.nbytes), _lib.DeviceBuffer(2 * img.nbytes)
        d1.upload(img)
        for _ in range(3):
            plan.extract_displacement_field_async(d1.ptr, kvecs, klists, sigma, 2 * sigma, 10, u1.ptr)
        plan.sync()
        t0 = time.perf_counter()
        for _ in range(4 * args.reps):
            plan.extract_displacement_field_async(d1.ptr, kvecs, klists, sigma, 2 * sigma, 10, u1.ptr)
        plan.sync()
        t1 = (time.perf_counter() - t0) / (4 * args.reps)
        print('%4d^2  single-image driver        %7.3f ms/image  %6.0f Mpix/s' % (n, t1 * 1e3, n * n / t1 / 1e6), flush=True)
        for B in (int(v) for v in args.stacks.split(',')):
            if B * img.nbytes * 40 > 60e9:
                continue
            stack = np.stack([img] * B)
            d, u = _lib.DeviceBuffer(stack.nbytes), _lib.DeviceBuffer(2 * stack.nbytes)
            d.upload(stack)
            for _ in range(2):
                plan.extract_displacement_field_batch_dev(d.ptr, B, kvecs, klists, sigma, 2 * sigma, 10, u.ptr, want_iters=False)
            plan.sync()
            t0 = time.perf_counter()
            for _ in range(args.reps):
                plan.extract_displacement_field_batch_dev(d.ptr, B, kvecs, klists, sigma, 2 * sigma, 10, u.ptr, want_iters=False)
            plan.sync()
            dt = (time.perf_counter() - t0) / args.reps
            print('%4d^2  stack of %3d               %7.3f ms/image  %6.0f Mpix/s  (x%.2f)' %
                  (n, B, dt / B * 1e3, n * n * B / dt / 1e6, t1 / (dt / B)), flush=True)
            d.free()
            u.free()
        d1.free()
        u1.free()
        plan.close()


if __name__ == '__main__':
    main()
