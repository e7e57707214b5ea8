#!/usr/bin/env python
"""Headline benchmark: Mpixels/s of displacement-field extraction on MI355X.

Workload (BASELINE.json config 3): 4096 x 4096 synthetic hex moire, 3 Bragg peaks x
16 reference k-vectors (explicit 4x4 lists), sigma = 10, weighted DCT-PCG unwrap
with kmax = 10, fp32.  One "step" = one whole extract_displacement_field call on an
image already resident in HBM (mean, 48 lock-ins, select, phases/weights, per-pixel
least squares, two unwraps), through the C ABI of libgpa_hip.so.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size S] [--dtype f32|f64]

N > 1 (launched by torch.distributed.run, one rank per GPU): every rank extracts
the displacement field of its own 4096^2 tile (weak scaling, no data-path collective
inside the extraction), then the tiles' fields are stitched with one RCCL
all_gather over xGMI inside the timed step.

Rank 0 prints ONE JSON line with the contract fields plus `roofline` (dominant
kernel, HIP-event timed on the plan's stream) and, at N = 1, `cpu_baseline` (the
NumPy/SciPy oracle timed on this box's host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 achievable)


def algorithmic_bytes(n0, n1, P, K, s, iters):
    """Algorithmic HBM bytes of one step and of each sweep kernel launch.

    pass A: read the real image once, write one complex intermediate per lock-in.
    pass B: read those intermediates, write P complex lock-ins.
    reconstruct: read P lock-ins, write dudx, dudy (2 comps each) and wnorm.
    unwrap (per component): setup 7s, 19s per PCG iteration."""
    px = n0 * n1
    B = P * K
    a = px * (s + 2 * s * B)
    b = px * (2 * s * B + 2 * s * P)
    rec = px * (2 * s * P + 5 * s)
    unw = sum(px * (7 * s + 19 * s * it) for it in iters)
    return {'passA': a, 'passB': b, 'reconstruct': rec, 'unwrap': unw, 'total': a + b + rec + unw}


def cpu_baseline(kvecs, sigma, knx, kny, kmax):
    """Time the CPU oracle (NumPy/SciPy port of the reference path) on bounded samples of the same
    workload (same P x K, same kmax), about 15-25 s of host work in total."""
    from oracle import gpa_oracle as orc
    from pygpa_amd.synthetic import gaussian_bump_displacement, hex_moire, explicit_klists
    cores = os.cpu_count() or 1
    kw = np.linalg.norm(kvecs, axis=1).mean() / 2.5
    klists = explicit_klists(kvecs, kw, knx, kny)
    # (i) reference-faithful threading: pyGPA runs single-threaded pocketfft and a serial per-pixel
    # solve (1024^2 sample); (ii) best effort: scipy.fft on every host core (2048^2 sample).
    # The faster rate is the baseline.
    runs = []
    for w, sample in ((1, 1024), (cores, 2048)) if cores > 1 else ((1, 1024),):
        shape = (sample, sample)
        img = hex_moire(shape, kvecs, gaussian_bump_displacement(shape), noise=0.1, seed=7)
        t = time.perf_counter()
        orc.extract_displacement_field(img, kvecs, sigma=sigma, klists=klists, workers=w)
        dt = time.perf_counter() - t
        runs.append((sample * sample / dt / 1e6, w, sample, dt))
    rate, w, sample, dt = max(runs)
    return {'value': round(rate, 4), 'unit': 'Mpixels/s', 'cores': w, 'kind': 'port',
            'sample': '%dx%d image, 3 peaks x %d k-vectors + weighted unwrap kmax=%d, oracle/gpa_oracle.py, scipy.fft '
                      'workers=%d: %.1f s; all runs: ' % (sample, sample, knx * kny, kmax, w, dt) +
                      ', '.join('%d^2 workers=%d %.1f s = %.3f Mpix/s' % (r[2], r[1], r[3], r[0]) for r in runs) +
                      ' (host has %d cores)' % cores}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--size', type=int, default=4096)
    ap.add_argument('--kside', type=int, default=4, help='k-vectors per peak = kside^2')
    ap.add_argument('--kgrid', default=None, help='NXxNY candidate grid per peak (e.g. 4x2 for BASELINE config 2)')
    ap.add_argument('--dtype', default='f32', choices=['f32', 'f64'])
    ap.add_argument('--kmax', type=int, default=10)
    ap.add_argument('--no-cpu', action='store_true', help='skip the CPU baseline leg')
    ap.add_argument('--force-torch', action='store_true', help='use the torch buffer path at N = 1 too (test aid)')
    ap.add_argument('--inflight', type=int, default=1,
                    help='images in flight per GPU (one plan each): 1 = strictly one after the other')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    torch = dist = None
    use_torch = world > 1 or args.force_torch
    if use_torch:
        # torch is plumbing for the multi-GPU run only (process group + RCCL); it is imported
        # BEFORE libgpa_hip.so so that both share the HIP runtime torch ships.  A single-GPU run
        # does not need it (and its first import on a cold machine takes minutes).
        import torch
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.cuda.set_device(local_rank)
        if world > 1:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))
    from pygpa_amd import _lib
    from pygpa_amd.synthetic import hex_kvecs, gaussian_bump_displacement, hex_moire, explicit_klists
    dev_index = local_rank if world > 1 else 0

    n = args.size
    knx, kny = (int(v) for v in args.kgrid.split('x')) if args.kgrid else (args.kside, args.kside)
    P, K = 3, knx * kny
    np_dt = np.float32 if args.dtype == 'f32' else np.float64
    s = 4 if args.dtype == 'f32' else 8
    kvecs = hex_kvecs(0.1, 7.0)
    sigma = int(np.ceil(1 / np.linalg.norm(kvecs, axis=1).min()))
    kw = np.linalg.norm(kvecs, axis=1).mean() / 2.5
    klists = np.stack(explicit_klists(kvecs, kw, knx, kny))

    # every rank owns one tile of a (world * n) x n synthetic image
    u_true = gaussian_bump_displacement((n, n))
    img = hex_moire((n, n), kvecs, u_true, noise=0.1, seed=100 + rank, dtype=np_dt)
    depth = max(1, args.inflight)
    # one plan (workspace + streams) per image in flight
    plans = [_lib.Plan((n, n), P * K, np_dt, device=dev_index) for _ in range(depth)]
    # u buffers: with N > 1 at least two, so that the all_gather of step i (RCCL, torch's stream) overlaps
    # the kernels of step i + 1 (plan streams), which write the other buffer
    nb = depth if world == 1 else max(2, depth)
    if use_torch:
        dev = torch.device('cuda', dev_index)
        t_dt = torch.float32 if args.dtype == 'f32' else torch.float64
        t_img = torch.from_numpy(img).to(dev)
        t_us = [torch.empty((2, n, n), dtype=t_dt, device=dev) for _ in range(nb)]
        d_alls = [torch.empty((world, 2, n, n), dtype=t_dt, device=dev) for _ in range(nb)] if world > 1 else None
        img_ptr, u_ptrs = t_img.data_ptr(), [t.data_ptr() for t in t_us]
    else:
        bufs = [_lib.DeviceBuffer(img.nbytes)] + [_lib.DeviceBuffer(2 * img.nbytes) for _ in range(nb)]
        bufs[0].upload(img)
        img_ptr, u_ptrs = bufs[0].ptr, [b.ptr for b in bufs[1:]]
        t_us = d_alls = None
    gathered = [None] * nb
    plan = plans[0]

    def gather(pj, j):
        # stitch the tiles' fields of the step that ran on plan pj into buffer j (RCCL all_gather over xGMI);
        # asynchronous on torch's stream, the event guards the reuse of u buffer j
        plans[pj].sync()
        dist.all_gather_into_tensor(d_alls[j], t_us[j])
        ev = torch.cuda.Event()
        ev.record()
        gathered[j] = ev

    def run(nsteps):
        pending = []
        for i in range(nsteps):
            j, pj = i % nb, i % depth
            if gathered[j] is not None:        # the collective that reads t_us[j] must be done before it is rewritten
                gathered[j].synchronize()
                gathered[j] = None
            plans[pj].extract_displacement_field_async(img_ptr, kvecs, klists, sigma, 2 * sigma, args.kmax, u_ptrs[j])
            if world > 1:
                pending.append((pj, j))
                if len(pending) > depth - 1:
                    gather(*pending.pop(0))
        for pj, j in pending:
            gather(pj, j)

    def fence():
        for pl in plans:
            pl.sync()
        if use_torch:
            torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    run(args.warmup)
    fence()
    t0 = time.perf_counter()
    run(args.steps)
    fence()
    dt = time.perf_counter() - t0
    iters = plan.last_iters()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=torch.device('cuda', dev_index))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # per-kernel timing of the dominant kernels, HIP events on the plan's own stream
    plan.set_profiling(True)
    stage = np.zeros(5)
    nprof = 5
    for _ in range(nprof):
        plan.extract_displacement_field_dev(img_ptr, kvecs, klists, sigma, 2 * sigma, args.kmax, u_ptrs[0])
        stage += np.array(plan.last_stage_ms())
    stage /= nprof
    plan.set_profiling(False)

    if rank == 0:
        ab = algorithmic_bytes(n, n, P, K, s, iters)
        names = ['tables+mean', 'passA_kernel', 'passB_kernel', 'reconstruct_kernel', 'unwrap(all kernels)']
        dom = 1 if stage[1] >= stage[2] else 2
        dom_bytes = ab['passA'] if dom == 1 else ab['passB']
        achieved = dom_bytes / (stage[dom] * 1e-3) / 1e9
        traffic = None
        tfile = os.path.join(ROOT, 'profiles', 'traffic.json')
        if os.path.exists(tfile):
            try:
                traffic = json.load(open(tfile)).get(names[dom])
            except Exception:
                traffic = None
        ms_per_step = dt / args.steps * 1e3
        out = {
            'metric': 'Mpixels/s displacement-field extraction (3 peaks, 4096^2 img) + achieved HBM GB/s',
            'value': round(world * n * n * args.steps / dt / 1e6, 2),
            'unit': 'Mpixels/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(ms_per_step, 4),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': '%dx%d synthetic hex moire per GPU, 3 Bragg peaks x %d k-vectors, sigma=%d, '
                                   'weighted DCT-PCG unwrap kmax=%d (BASELINE.json configs[2])' % (n, n, K, sigma, args.kmax),
                       'image': [n, n], 'peaks': P, 'kvectors_per_peak': K, 'unwrap_iters': list(iters),
                       'sharding': 'one image tile per rank' + (', RCCL all_gather of u' if world > 1 else ''),
                       'images_in_flight_per_gpu': depth},
            'algorithmic_GBps_whole_step': round(ab['total'] / (ms_per_step * 1e-3) / 1e9, 1),
            'stage_ms': {names[i]: round(float(stage[i]), 4) for i in range(5)},
            'roofline': {'bound': 'hbm', 'kernel': names[dom], 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS,
                         'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': traffic,
                         'algorithmic_bytes_per_launch': dom_bytes, 'kernel_ms': round(float(stage[dom]), 4)},
        }
        if world == 1 and not args.no_cpu:
            out['cpu_baseline'] = cpu_baseline(kvecs, sigma, knx, kny, args.kmax)
        print(json.dumps(out), flush=True)
    for pl in plans:
        pl.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
