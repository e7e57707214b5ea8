#!/usr/bin/env python3
"""Shared-forward pass B against the per-candidate pass B (GPA_NO_SHARED=1, run in a child process) and the
oracle: sweep of one peak on square / rectangular images, both precisions; prints max deviations, the agreement of
the winner index and, separately, the deviations in the first / last 3 sigma columns (where the end fix acts).

    python tools/check_shared_passb.py [--sizes 1024,2048] [--oracle]
"""
import argparse
import os
import subprocess
import sys
import pickle

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run_case(shape, dtype, kx, ky):
    from pygpa_amd import _lib
    from pygpa_amd.synthetic import hex_kvecs, gaussian_bump_displacement, hex_moire, explicit_klists
    kvecs = hex_kvecs(0.1, 7.0)
    img = hex_moire(shape, kvecs, gaussian_bump_displacement(shape), noise=0.2, seed=11)
    img0 = img - img.mean()
    from oracle import gpa_oracle as orc
    kw, sigma, _ = orc.derive_params(kvecs)
    klist = explicit_klists(kvecs, kw, kx, ky)[1]
    plan = _lib.Plan(shape, len(klist), dtype)
    lock, kidx, _ = plan.sweep(img0, kvecs[1], klist, sigma)
    plan.close()
    return img0, sigma, klist, kvecs[1], lock, kidx


def child(args):
    shape = tuple(int(v) for v in args.child.split('x'))
    out = {}
    for dt in (np.float32, np.float64):
        _, _, _, _, lock, kidx = run_case(shape, dt, args.kx, args.ky)
        out[np.dtype(dt).name] = (lock, kidx)
    pickle.dump(out, sys.stdout.buffer)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--sizes', default='1024x1024,2048x1024,1024x4096')
    ap.add_argument('--oracle', action='store_true')
    ap.add_argument('--child', default=None)
    ap.add_argument('--kx', type=int, default=4)
    ap.add_argument('--ky', type=int, default=4)
    args = ap.parse_args()
    if args.child:
        return child(args)
    for sz in args.sizes.split(','):
        shape = tuple(int(v) for v in sz.split('x'))
        env = dict(os.environ, GPA_NO_SHARED='1')
        old = pickle.loads(subprocess.run([sys.executable, __file__, '--child', sz, '--kx', str(args.kx), '--ky', str(args.ky)],
                                          env=env, stdout=subprocess.PIPE, check=True).stdout)
        for dt in (np.float32, np.float64):
            img0, sigma, klist, kref, lock, kidx = run_case(shape, dt, args.kx, args.ky)
            lo, ko = old[np.dtype(dt).name]
            sc = np.abs(lo).max()
            same = kidx == ko
            e3 = int(3 * sigma)
            d = np.abs(lock - lo)
            dm = np.where(same, d, 0)
            print('%-10s %s  kidx equal %.6f  |new-old| max %.3e (same-winner px)  ends(3 sigma) %.3e  interior %.3e' % (
                sz, np.dtype(dt).name, same.mean(), dm.max() / sc, max(dm[:, :e3].max(), dm[:, -e3:].max()) / sc,
                dm[:, e3:-e3].max() / sc), flush=True)
            if args.oracle:
                from oracle import gpa_oracle as orc
                ref = orc.sweep(img0, sigma, klist, kref, workers=8)
                s2 = kidx == ref['kidx']
                dr = np.where(s2, np.abs(lock - ref['lockin']), 0)
                s3 = ko == ref['kidx']
                dro = np.where(s3, np.abs(lo - ref['lockin']), 0)
                print('           vs oracle: kidx equal %.6f (old %.6f)  max %.3e (old %.3e)  ends %.3e (old %.3e)' % (
                    s2.mean(), s3.mean(), dr.max() / sc, dro.max() / sc, max(dr[:, :e3].max(), dr[:, -e3:].max()) / sc,
                    max(dro[:, :e3].max(), dro[:, -e3:].max()) / sc), flush=True)


if __name__ == '__main__':
    main()
