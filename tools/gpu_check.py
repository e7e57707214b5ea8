#!/usr/bin/env python
"""Diagnostic parity run on a GPU box: prints error figures for every stage instead
of asserting, so one gpurun call tells where a new build stands."""
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import gpa_oracle as orc          # noqa: E402
from pygpa_amd import _lib                     # noqa: E402
from pygpa_amd.synthetic import hex_kvecs, gaussian_bump_displacement, hex_moire, explicit_klists  # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')


def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def stage(name, fn):
    t = time.time()
    try:
        fn()
    except Exception:
        print('[%s] EXCEPTION' % name)
        traceback.print_exc()
    print('[%s] %.2fs' % (name, time.time() - t), flush=True)


def check_case(name, dtype):
    g = dict(np.load(os.path.join(GOLD, name + '.npz')))
    img = g['image']
    img0 = img - img.mean()
    sigma = int(g['sigma'])
    kvecs = g['kvecs']
    K = g['a3_klists'].shape[1]
    plan = _lib.Plan(img.shape, 3 * K, dtype)
    tag = '%s/%s' % (name, np.dtype(dtype).name)
    print('%s: fft lens %d x %d, workspace %.1f MB' % (tag, plan.fft_len(0), plan.fft_len(1), plan.workspace_bytes / 1e6))
    if 'a1_GPA' in g:
        out = plan.lockin_batch(img0, kvecs[:2], sigma)
        print('  a1 lockin rel err  %.3e %.3e' % (rel(out[0], g['a1_GPA']), rel(out[1], g['a1_optGPA'])))
    for p in range(3):
        lock, kidx, _ = plan.sweep(img0, kvecs[p], g['a3_klists'][p], sigma)
        ref = g['a3_lockin'][p] if 'a3_lockin' in g else (g['a3_lockin0'] if p == 0 else None)
        mism = int((kidx != g['a3_kidx'][p]).sum())
        msg = '  a3 sweep peak %d: kidx mismatches %d / %d' % (p, mism, kidx.size)
        if ref is not None:
            same = kidx == g['a3_kidx'][p]
            msg += ', lockin rel err (matching px) %.3e' % rel(lock[same], ref[same])
        print(msg)
    if 'a3_lockin' in g:
        dudx, dudy, wn = plan.reconstruct_grad(g['a3_lockin'], kvecs, 2 * sigma)
        print('  a6 dudx/dudy rel err %.3e %.3e  wnorm %.3e' % (rel(dudx, g['a6_dudx']), rel(dudy, g['a6_dudy']),
                                                              rel(wn, np.linalg.norm(g['a5_weights'], axis=0))))
        wn64 = np.linalg.norm(g['a5_weights'], axis=0)
        for kmax in (1, 3, 10, 100):
            try:
                phi, it = plan.unwrap_prediff(g['a6_dudx'][0], g['a6_dudy'][0], wn64, kmax=kmax)
                print('  a7 unwrap kmax=%3d iters=%3d rel err %.3e' % (kmax, it, rel(phi, g['a7_phi_w_kmax%d' % kmax])))
            except _lib.GPAError as e:
                print('  a7 unwrap: %s' % e)
                break
        try:
            phi, it = plan.unwrap_prediff(g['a6_dudx'][0], g['a6_dudy'][0])
            print('  a7 unweighted iters=%d rel err %.3e' % (it, rel(phi, g['a7_phi_unweighted'])))
        except _lib.GPAError as e:
            print('  a7 unweighted: %s' % e)
    try:
        u, lock, kidx, iters = plan.extract_displacement_field(img, kvecs, g['a3_klists'], sigma, 2 * sigma, kmax=10,
                                                                want_lockins=True, want_kidx=True)
        print('  driver: u rel err %.3e  kidx mismatches %d iters %s' % (rel(u, g['u']), int((kidx != g['a3_kidx']).sum()), iters))
    except _lib.GPAError as e:
        print('  driver: %s' % e)
    plan.close()


def check_large(n, dtype, K2=2):
    kvecs = hex_kvecs()
    u_true = gaussian_bump_displacement((n, n))
    img = hex_moire((n, n), kvecs, u_true, noise=0.1, seed=5)
    img0 = img - img.mean()
    plan = _lib.Plan((n, n), 4, dtype)
    ks = np.array([kvecs[0], kvecs[1] + 0.013])
    t = time.time()
    out = plan.lockin_batch(img0, ks, 10)
    tg = time.time() - t
    t = time.time()
    ref = orc.lockin_batch(img0, ks, 10, workers=8)
    print('large %d %s: lockin rel err %.3e (gpu call %.2fs incl. copies, oracle %.2fs)' %
          (n, np.dtype(dtype).name, rel(out, ref), tg, time.time() - t))
    plan.close()


def main():
    lib = _lib.load()
    print('gpa_version', lib.gpa_version(), 'devices', lib.gpa_device_count())
    for dt in (np.float64, np.float32):
        for name in ('hex_64', 'hex_48x80', 'hex_63x65', 'hex_128_noise'):
            stage(name, lambda: check_case(name, dt))
    for dt in (np.float64, np.float32):
        stage('large512', lambda: check_large(512, dt))
        stage('large2048', lambda: check_large(2048, dt))
    stage('large4096', lambda: check_large(4096, np.float32))


if __name__ == '__main__':
    main()
