"""Throughput on a STACK of small images: D plans, each driven by its own host thread (ctypes releases the GIL, so
the ~110 launches per image are enqueued in parallel), images left in HBM.
    python tools/stack_throughput.py SIZE [--depths 1,2,4,8] [--images 64]"""
import argparse
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygpa_amd import _lib                                                     # noqa: E402
from pygpa_amd.synthetic import explicit_klists, gaussian_bump_displacement, hex_kvecs, hex_moire   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('size', type=int)
    ap.add_argument('--depths', default='1,2,4,8')
    ap.add_argument('--images', type=int, default=64)
    ap.add_argument('--ahead', type=int, default=2, help='images a thread enqueues before it waits for its plan')
    args = ap.parse_args()
    n = args.size
    kvecs = hex_kvecs(0.1, 7.0)
    sigma = int(np.ceil(1 / np.linalg.norm(kvecs, axis=1).min()))
    kw = np.linalg.norm(kvecs, axis=1).mean() / 2.5
    klists = np.stack(explicit_klists(kvecs, kw, 4, 4))
    img = hex_moire((n, n), kvecs, gaussian_bump_displacement((n, n)), noise=0.1, seed=1, dtype=np.float32)
    d_img = _lib.DeviceBuffer(img.nbytes)
    d_img.upload(img)
    for depth in (int(v) for v in args.depths.split(',')):
        plans = [_lib.Plan((n, n), 48, np.float32, device=0) for _ in range(depth)]
        bufs = [[_lib.DeviceBuffer(2 * img.nbytes) for _ in range(args.ahead)] for _ in range(depth)]

        def work(t, count):
            p = plans[t]
            for i in range(count):
                p.extract_displacement_field_async(d_img.ptr, kvecs, klists, sigma, 2 * sigma, 10, bufs[t][i % args.ahead].ptr)
                if i % args.ahead == args.ahead - 1:
                    p.sync()
            p.sync()

        for mode in ('threads', 'serial'):
            if mode == 'serial' and depth == 1:
                continue
            per = args.images // depth
            # warm-up
            ths = [threading.Thread(target=work, args=(t, 2)) for t in range(depth)]
            [t.start() for t in ths]
            [t.join() for t in ths]
            t0 = time.perf_counter()
            if mode == 'threads':
                ths = [threading.Thread(target=work, args=(t, per)) for t in range(depth)]
                [t.start() for t in ths]
                [t.join() for t in ths]
            else:   # one host thread feeding all plans round robin
                for i in range(per):
                    for t in range(depth):
                        plans[t].extract_displacement_field_async(d_img.ptr, kvecs, klists, sigma, 2 * sigma, 10,
                                                                  bufs[t][i % args.ahead].ptr)
                    if i % args.ahead == args.ahead - 1:
                        [p.sync() for p in plans]
                [p.sync() for p in plans]
            dt = time.perf_counter() - t0
            print('%d^2  D=%d  %-7s  %.3f ms/image  %.0f Mpix/s' % (n, depth, mode, dt / (per * depth) * 1e3, n * n * per * depth / dt / 1e6), flush=True)
        for p in plans:
            p.close()
        for pair in bufs:
            for b in pair:
                b.free()


if __name__ == '__main__':
    main()
