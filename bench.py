#!/usr/bin/env python
"""Headline benchmark: Mpixels/s of displacement-field extraction on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size S] [--dtype f32|f64]

N = 1 (BASELINE.json configs[2]): 4096 x 4096 synthetic hex moire, 3 Bragg peaks x 16 reference
k-vectors (explicit 4x4 lists), sigma = 10, weighted DCT-PCG unwrap with kmax = 10, fp32.  One step =
one whole extract_displacement_field call through the C ABI of libgpa_hip.so on an image already
resident in HBM (mean, 48 lock-ins, select, phases/weights, per-pixel least squares, two unwraps)
PLUS the download of u to page-locked host memory ("D2H of u included"), which runs on
the plan's copy stream while the kernels of the next step execute.  The PCG of the timed step stops by the
reference's test alone (kmax or ||r|| < 1e-9 ||r0||: 10 + 10 iterations on this image) -- the library default; the
opt-in f32 residual floor (F32_EPS_FLOOR=4e-6) is the extra key `early_stop`.  Other extra keys: `resident_only`
(the same loop with u left in HBM), `f64` (the reference's own precision), `host_call` (host arrays in and out: H2D
and D2H inside the call), `kernels` (per-kernel HIP-event times with the roofline that bounds each), `cpu_baseline`.

N > 1 (one rank per GPU; `--gpus N` without a launcher starts N fresh child processes through
torch.distributed.run before this process touches a GPU): the tile pipeline of BASELINE configs[3-4],
pygpa_amd.distributed.TiledPipeline, on an image of N x 4096^2 pixels (N = 4: 8192^2 = configs[3]):
halo windows dealt over the ranks -> all_gather of the gradient tiles -> global unwrap -> broadcast of u.
Weak scaling: pixels per GPU are fixed.

Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 achievable)
VALU_PEAK_TFLOPS = 157.3     # f32 vector peak: 1024 SIMD-32 x 2 flop x 2.4 GHz (same guide)
VALU_PEAK_TFLOPS_F64 = 78.6  # f64 vector peak (same guide): half the f32 rate
VALU_ISSUE_PEAK = 1024 * 2.4e9 / 2   # wave64 VALU instructions / s: one per 2 cycles per SIMD


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    # (defaults: the timed loop ends with the pipeline's drain -- the last image's D2H of u, 2.6 ms, overlaps nothing -- which
    #  100 steps amortise to 0.5 % of a step where 20 steps left 2.4 %: tools/ab_steps.sh; the whole default run stays under a minute)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--size', type=int, default=4096)
    ap.add_argument('--kside', type=int, default=4, help='k-vectors per peak = kside^2')
    ap.add_argument('--kgrid', default=None, help='NXxNY candidate grid per peak (e.g. 4x2 for BASELINE config 2)')
    ap.add_argument('--dtype', default='f32', choices=['f32', 'f64'])
    ap.add_argument('--kmax', type=int, default=10)
    ap.add_argument('--inflight', type=int, default=1,
                    help='N = 1: images in flight (one plan each; 1 = strictly one after the other). Small images leave most '
                         'of the GPU idle between their ~45 dependent kernels; several in flight fill it')
    ap.add_argument('--no-cpu', action='store_true', help='skip the CPU baseline leg')
    ap.add_argument('--no-f64', action='store_true', help='skip the f64 leg')
    ap.add_argument('--no-lf', action='store_true', help='N > 1: skip the tile-sharded Lawler-Fujita diagnostic after the timed region')
    ap.add_argument('--no-pipeline', action='store_true', help='skip the image -> k-vectors -> u -> undistortion -> properties leg (pipeline_end_to_end)')
    ap.add_argument('--only-pipeline', action='store_true', help='run that leg alone and print its object (tools / profiles)')
    ap.add_argument('--no-config5', action='store_true', help='skip the 16384^2 tile pipeline + Lawler-Fujita leg (config5_single_gpu)')
    ap.add_argument('--window', type=int, default=2048, help='N > 1: side of the (power-of-two) tile windows')
    ap.add_argument('--backend', default='nccl', help='N > 1: torch.distributed backend (gloo stages through the host)')
    ap.add_argument('--share-device', action='store_true', help='N > 1: every rank uses GPU 0 (test aid, with --backend gloo)')
    ap.add_argument('--schedule', default='stream', choices=['stream', 'step'],
                    help='N > 1: image-pipelined schedule (TiledPipeline.run_stream) or the unpipelined step()')
    return ap.parse_args()


# -------------------------------------------------------------------------------------------------
# byte / flop models
# -------------------------------------------------------------------------------------------------
def survey_bytes(n0, n1, P, K, s, iters):
    """SURVEY.md 8(d)'s algorithmic bytes of one step (a 2-D FFT pair per lock-in, 19 arrays per PCG
    iteration): the reference algorithm's traffic, kept as a yardstick."""
    px = n0 * n1
    sweep = px * P * (8 * s * K + 3 * s)
    rec = px * (2 * s * P + 5 * s)
    unw = sum(px * (7 * s + 19 * s * it) for it in iters)
    return {'sweep': sweep, 'reconstruct': rec, 'unwrap': unw, 'total': sweep + rec + unw}


def kernel_models(n0, n1, L0, L1, P, K, Bx, s, iters):
    """Per-launch algorithmic HBM bytes (what THIS build's kernels must move: every operand once) and
    nominal flops (5 L log2 L per complex FFT of length L) of each kernel of the step."""
    px = n0 * n1
    B = P * K
    fft = lambda L: 5.0 * L * np.log2(L)
    return {
        # image once, one complex x-plane per distinct wx out
        'passA_kernel': {'bytes': px * (s + 2 * s * Bx), 'flops': Bx * n1 * (2 * fft(L0) + 8 * L0)},
        # every x-plane in once (the K/Bx-fold re-reads of a plane are served by L2), P winners out
        'passB_kernel': {'bytes': px * (2 * s * Bx + 2 * s * P), 'flops': B * n0 * (2 * fft(L1) + 16 * L1)},
        # the same traffic and the same ALGORITHMIC flops (the reference's lock-in of every candidate: a forward and an
        # inverse transform per candidate and row); the kernel EXECUTES one forward transform per x-plane
        # row (Bx) and one inverse per candidate (B) -- 'executed_flops'
        'passB_shared_kernel': {'bytes': px * (2 * s * Bx + 2 * s * P), 'flops': B * n0 * (2 * fft(L1) + 16 * L1),
                                'executed_flops': n0 * ((Bx + B) * fft(L1) + B * 10 * L1)},
        # P lock-ins in; wnorm, r0 of both components out
        'reconstruct_setup_kernel': {'bytes': px * (2 * s * P + 3 * s), 'flops': None},
        # per working launch and component: q, R in, R out / R in, Z out / Z, p in, p out / p, w in, q out
        'rowdct_fused_kernel': {'bytes': px * 3 * s, 'flops': (n0 / 2) * (fft(L1) + 12 * L1)},
        'colsolve_kernel': {'bytes': px * 2 * s, 'flops': (n1 / 2) * (2 * fft(L0) + 24 * L0)},
        'colsolve_tri_kernel': {'bytes': px * 2 * s, 'flops': None},
        # the streamed column solve: R in (chunk sums out: 16 B per 64 samples) / sums in, carries out / R + carries in, Z out
        'colstream_agg_kernel': {'bytes': px * s, 'flops': None},   # (R in; with the stencil-fused iteration R, D in, R out: kernel_table)
        # stencil + row transform in one launch: p, w in, D = DCT_rows(q) out
        'pqdct_kernel': {'bytes': px * 3 * s, 'flops': (n0 / 2) * (fft(L1) + 12 * L1)},
        'colstream_scan_kernel': {'bytes': None, 'flops': None},
        'colstream_apply_kernel': {'bytes': px * 2 * s, 'flops': None},
        'rowidct_p_kernel': {'bytes': px * 3 * s, 'flops': (n0 / 2) * (fft(L1) + 12 * L1)},
        'pq_kernel': {'bytes': px * 3 * s, 'flops': None},
        # phi in/out once per flush plus the kept search directions
        'phi_flush_kernel': {'bytes': None, 'flops': None},
    }


def cpu_baseline(kvecs, sigma, knx, kny, kmax, size):
    """Time the CPU oracle (NumPy/SciPy port of the reference path) on the GPU box's host cores:
    (i) reference-faithful threading (single-threaded pocketfft, serial per-pixel solve) on a 1024^2
    sample; (ii) best effort on every core -- the candidates of a peak through a thread pool (the
    analogue of the reference's dask wfr2_only_lockin_vec) and scipy.fft workers for the unwrap -- on
    the benchmark's own image size when that fits the time budget, else 2048^2."""
    from oracle import gpa_oracle as orc
    from pygpa_amd.synthetic import gaussian_bump_displacement, hex_moire, explicit_klists
    cores = os.cpu_count() or 1
    kw = np.linalg.norm(kvecs, axis=1).mean() / 2.5
    klists = explicit_klists(kvecs, kw, knx, kny)

    def run(sample, workers, pool):
        shape = (sample, sample)
        img = hex_moire(shape, kvecs, gaussian_bump_displacement(shape), noise=0.1, seed=7)
        t = time.perf_counter()
        orc.extract_displacement_field(img, kvecs, sigma=sigma, klists=klists, workers=workers, pool=pool)
        dt = time.perf_counter() - t
        return {'rate': sample * sample / dt / 1e6, 'cores': max(workers, pool), 'sample': sample, 'seconds': dt,
                'mode': 'pool=%d workers=%d' % (pool, workers)}

    runs = [run(1024, 1, 1)]
    if cores > 1:
        pool = min(cores, knx * kny)
        probe = run(1024, cores, pool)
        runs.append(probe)
        # the multi-core run scales ~ with pixels: take the benchmark size if it stays under ~45 s
        full = size if probe['seconds'] * (size / 1024.0) ** 2 < 45.0 else 2048
        if full > 1024:
            runs.append(run(full, cores, pool))
    best = max(runs, key=lambda r: r['rate'])      # the GPU/CPU ratio is quoted against the fastest CPU run
    return {'value': round(best['rate'], 4), 'unit': 'Mpixels/s', 'cores': best['cores'], 'kind': 'port',
            'sample': '%dx%d image, 3 peaks x %d k-vectors + weighted unwrap kmax=%d, oracle/gpa_oracle.py (%s): %.1f s; '
                      'all runs: ' % (best['sample'], best['sample'], knx * kny, kmax, best['mode'], best['seconds']) +
                      ', '.join('%d^2 %s %.1f s = %.3f Mpix/s' % (r['sample'], r['mode'], r['seconds'], r['rate']) for r in runs) +
                      ' (host has %d cores)' % cores}


# -------------------------------------------------------------------------------------------------
# N = 1
# -------------------------------------------------------------------------------------------------
class SingleGPU:
    def __init__(self, n, P, K, np_dt, kvecs, klists, sigma, kmax, seed=100, depth=1):
        from pygpa_amd import _lib
        from pygpa_amd.synthetic import gaussian_bump_displacement, hex_moire
        self._lib = _lib
        self.n, self.kvecs, self.klists, self.sigma, self.kmax = n, kvecs, klists, sigma, kmax
        img = hex_moire((n, n), kvecs, gaussian_bump_displacement((n, n)), noise=0.1, seed=seed, dtype=np_dt)
        self.depth = max(1, int(depth))
        self.plans = [_lib.Plan((n, n), P * K, np_dt, device=0) for _ in range(self.depth)]
        self.plan = self.plans[0]
        self.d_img = _lib.DeviceBuffer(img.nbytes)
        self.d_img.upload(img)
        # per plan: two result buffers on the device and two page-locked ones on the host
        self.d_u = [[_lib.DeviceBuffer(2 * img.nbytes) for _ in range(2)] for _ in range(self.depth)]
        self.h_u = [[_lib.pinned_empty((2, n, n), np_dt) for _ in range(2)] for _ in range(self.depth)]
        for pl in range(self.depth):        # set-up, not benchmark steps: lazy allocations happen in the first calls
            for j in (0, 1):
                self.enqueue(pl, j)
        self.sync()

    def enqueue(self, pl, j):
        self.plans[pl].extract_displacement_field_async(self.d_img.ptr, self.kvecs, self.klists, self.sigma, 2 * self.sigma,
                                                        self.kmax, self.d_u[pl][j].ptr)

    def sync(self):
        for p in self.plans:
            p.sync()

    def run(self, nsteps, download):
        """nsteps steps; step i runs on plan i % depth.  With `download` the u of a step goes to pinned host memory on
        its plan's copy stream while later steps compute (into the plan's other buffer / on the other plans)"""
        for i in range(nsteps):
            pl, j = i % self.depth, (i // self.depth) & 1
            if download:
                self.plans[pl].download_wait(j)      # the copy that last read d_u[pl][j] has landed
            self.enqueue(pl, j)
            if download:
                self.plans[pl].download_async(self.h_u[pl][j], self.d_u[pl][j].ptr, j)
        self.sync()

    def timed(self, steps, warmup, download):
        self.run(warmup, download)
        t0 = time.perf_counter()
        self.run(steps, download)
        return time.perf_counter() - t0

    def profile(self, reps=3):
        self.plan.set_profiling(True)
        stage = np.zeros(5)
        kern = {}
        for _ in range(reps):
            self.plan.extract_displacement_field_dev(self.d_img.ptr, self.kvecs, self.klists, self.sigma, 2 * self.sigma,
                                                     self.kmax, self.d_u[0][0].ptr)
            stage += np.array(self.plan.last_stage_ms())
            for name, (calls, ms) in self.plan.last_kernel_profile().items():
                c, t = kern.get(name, (0, 0.0))
                kern[name] = (calls, t + ms)
        self.plan.set_profiling(False)
        return stage / reps, {k: (c, t / reps) for k, (c, t) in kern.items()}

    def close(self):
        for p in self.plans:
            p.close()
        for b in [self.d_img] + [x for pair in self.d_u for x in pair]:
            b.free()


def csrc_sha256():
    """hash of the kernel sources the counters were measured on (the GPU box has no .git: the tree itself is the
    witness).  tools/make_counters.py stamps it into profiles/counters.json."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, 'pygpa_amd', 'csrc', '*.h*'))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, 'rb').read())
    return h.hexdigest()[:16]


def load_counters(name='counters.json'):
    """per-launch PMC figures of the committed rocprofv3 passes (profiles/counters.json; the f64 leg: counters_f64.json,
    FETCH_SIZE / WRITE_SIZE only): HBM bytes (FETCH_SIZE doubled per the microarch guide + WRITE_SIZE) and VALU
    wave-instructions per kernel.
    REFUSED (empty: traffic null) when the file was measured on other kernel sources than this tree's."""
    path = os.path.join(ROOT, 'profiles', name)
    try:
        c = json.load(open(path))
    except Exception:
        return {}
    if c.get('_meta', {}).get('csrc_sha256') != csrc_sha256():
        return {'_stale': 'profiles/counters.json (commit %s) was measured on other kernel sources than this tree (csrc hash %s '
                          'vs %s): not used' % (c.get('_meta', {}).get('commit', '?'), c.get('_meta', {}).get('csrc_sha256', 'none'),
                                                csrc_sha256())}
    return c


def measure(n, knx, kny, np_dt, kmax, steps, warmup, depth=1, profile=True):
    """One configuration on GPU 0: timed loop with the D2H of u, the same loop with u left in HBM, and (profile) the
    per-kernel HIP-event table with the roofline of the dominant kernel."""
    from pygpa_amd.synthetic import hex_kvecs, explicit_klists
    P, K = 3, knx * kny
    kvecs = hex_kvecs(0.1, 7.0)
    sigma = int(np.ceil(1 / np.linalg.norm(kvecs, axis=1).min()))
    kw = np.linalg.norm(kvecs, axis=1).mean() / 2.5
    klists = np.stack(explicit_klists(kvecs, kw, knx, kny))
    Bx = sum(len(np.unique(kl[:, 0])) for kl in klists)
    s = np.dtype(np_dt).itemsize
    from pygpa_amd import _lib
    g = SingleGPU(n, P, K, np_dt, kvecs, klists, sigma, kmax, depth=depth)
    # The timed step runs the REFERENCE's stopping test alone (phase_unwrap.py:348: k >= kmax or ||r|| < 1e-9 ||r0||) --
    # the library default in both precisions since round 5.  The opt-in f32 residual floor (F32_EPS_FLOOR=4e-6)
    # ends the benchmark image's solves after 9 + 8 iterations instead of 10 + 10; it is reported as `early_stop`.
    dt = g.timed(steps, warmup, download=True)
    iters = g.plan.last_iters()
    dt_res = g.timed(steps, 1, download=False)
    res = {'n': n, 'P': P, 'K': K, 'Bx': int(Bx), 'sigma': sigma, 'kvecs': kvecs, 'klists': klists, 'iters': list(iters),
           'value': round(n * n * steps / dt / 1e6, 2), 'ms_per_step': round(dt / steps * 1e3, 4),
           'resident_value': round(n * n * steps / dt_res / 1e6, 2), 'resident_ms': round(dt_res / steps * 1e3, 4),
           'depth': g.depth}
    if profile:
        stage, kern = g.profile()
        L0, L1 = g.plan.fft_len(0), g.plan.fft_len(1)
        res.update(kernel_table(n, L0, L1, P, K, Bx, s, iters, kern, stage, dt_res / steps))
    if np_dt is np.float32:
        _lib.set_option('F32_EPS_FLOOR', '4e-6')
        try:
            dte = g.timed(steps, 1, download=True)
            res['early_stop'] = {'value': round(n * n * steps / dte / 1e6, 2), 'ms_per_step': round(dte / steps * 1e3, 4),
                                 'unwrap_iters': list(g.plan.last_iters()),
                                 'note': 'opt-in F32_EPS_FLOOR=4e-6: f32 solves also stop at that relative residual (not in the '
                                         'reference); same step, D2H of u included.  NOT the headline, NOT the library default.'}
        finally:
            _lib.set_option('F32_EPS_FLOOR', None)
    g.close()
    return res


def kernel_table(n, L0, L1, P, K, Bx, s, iters, kern, stage, res_s):
    """per-kernel table: HIP-event time (serial run of the step), the bound the counters show, fractions of peak; the
    roofline object of the dominant kernel; whole-step traffic figures"""
    out = {}
    models = kernel_models(n, n, L0, L1, P, K, Bx, s, iters)
    valu_peak = VALU_PEAK_TFLOPS if s == 4 else VALU_PEAK_TFLOPS_F64     # the vector peak of the precision the path computes in
    counters = load_counters('counters.json' if s == 4 else 'counters_f64.json')
    stale = counters.pop('_stale', None)
    meta = counters.get('_meta', {})
    same_cfg = meta.get('config', {'n': 4096, 'K': 16, 'dtype': 'f32'}) == {'n': n, 'K': K, 'dtype': 'f32' if s == 4 else 'f64'}
    if not same_cfg:
        counters = {}      # the committed counters belong to ONE configuration (their _meta says which)
    work = {k: sum(iters) for k in ('rowdct_fused_kernel', 'rowidct_p_kernel', 'pq_kernel', 'rowidct_pq_kernel', 'colsolve_kernel',
                                     'colsolve_tri_kernel', 'colstream_agg_kernel', 'colstream_scan_kernel', 'colstream_apply_kernel', 'pqdct_kernel')}   # launches that do work (those after convergence return at once)
    if 'pqdct_kernel' in kern and 'colstream_agg_kernel' in kern:
        # the stencil-fused iteration: the first launch of each component only sums (R in), the others apply R -= alpha D
        # as well (R, D in, R out): bytes per launch averaged over the launches of the step
        calls = kern['colstream_agg_kernel'][0]
        models['colstream_agg_kernel'] = {'bytes': n * n * s * (2 * 1 + 3 * max(calls - 2, 0)) / max(calls, 1), 'flops': None}
        # ... and the row transform of r0 / the plain stencil run once per component (first / last iteration), the fused
        # stencil + transform in every iteration between
        ncomp = len(iters)
        work.update({'rowdct_fused_kernel': ncomp, 'pq_kernel': ncomp, 'pqdct_kernel': max(sum(iters) - ncomp, 1)})
    table = {}
    for name, (calls, ms) in kern.items():
        m = models.get(name, {})
        c = counters.get(name, {})
        working = work.get(name, calls)
        row = {'launches': calls, 'working_launches': working, 'total_ms': round(ms, 4),
               'avg_us_all_launches': round(ms / max(calls, 1) * 1e3, 2)}
        per = ms * 1e-3 / max(working, 1)          # seconds per working launch (early-exit launches take ~2 us)
        if m.get('bytes'):
            row['algorithmic_GBps'] = round(m['bytes'] / per / 1e9, 1)
            row['algorithmic_bytes_per_launch'] = int(m['bytes'])
        if c.get('hbm_bytes'):
            row['hbm_GBps'] = round(c['hbm_bytes'] / per / 1e9, 1)
            row['hbm_frac'] = round(c['hbm_bytes'] / per / 1e9 / HBM_PEAK_GBS, 4)
            row['hbm_bytes_per_launch'] = int(c['hbm_bytes'])
        if m.get('flops'):
            row['nominal_TFLOPs'] = round(m['flops'] / per / 1e12, 2)
        if m.get('executed_flops'):
            row['executed_TFLOPs'] = round(m['executed_flops'] / per / 1e12, 2)
        if c.get('valu_insts'):
            row['valu_issue_frac'] = round(c['valu_insts'] / per / VALU_ISSUE_PEAK, 4)
            # the same count priced at what a SIMD with 2-3 resident waves retires (one wave64 instruction per ~4 cycles; the
            # 2-cycle peak needs 8 waves: tools/ubench/pkfma.hip 7.2 / 4.1 / 3.2 / 2.8 cycles at 1 / 2 / 4 / 8 waves)
            row['valu_busy_frac_4cycle'] = round(c['valu_insts'] * 4 / (1024 * 2.4e9) / per, 4)
        fr = {'hbm': row.get('hbm_frac', (m['bytes'] / per / 1e9 / HBM_PEAK_GBS) if m.get('bytes') else 0.0),
              'valu': row.get('valu_issue_frac', ((m.get('executed_flops') or m['flops']) / per / 1e12 / valu_peak) if m.get('flops') else 0.0)}
        row['bound'] = max(fr, key=fr.get)
        table[name] = row
    out['kernels'] = table
    out['stage_ms'] = {nm: round(float(v), 4) for nm, v in zip(
        ['tables+mean', 'passA_kernel', 'passB_kernel', 'reconstruct_setup', 'unwrap(serial, both components)'], stage)}

    # ---- roofline of the dominant kernel (largest single launch): pass B, VALU-issue bound
    dom = max((k for k in table if table[k]['launches'] <= 2 and (models.get(k, {}).get('bytes') or models.get(k, {}).get('flops'))),
              key=lambda k: table[k]['total_ms'], default='passB_shared_kernel')
    drow, dm, dc = table[dom], models.get(dom, {}), counters.get(dom, {})
    dsec = drow['total_ms'] * 1e-3 / max(drow['working_launches'], 1)
    src = {'clock': 'HIP events on the launch stream around every launch of the kernel, averaged over 3 profiled steps of '
                    'this run (adds ~3 us per launch to what rocprofv3 --kernel-trace reports: profiles/)',
           'counters': ('profiles/counters.json measured on commit %s, %s (%s)' % (meta.get('commit', '?'), meta.get('date', '?'),
                                                                                meta.get('how', 'rocprofv3 --pmc passes')))
                       if counters else (stale or 'none for this configuration (traffic: null)')}
    if drow['bound'] == 'valu' and dm.get('flops'):
        ex = dm.get('executed_flops', dm['flops'])
        out['roofline'] = {'bound': 'valu', 'kernel': dom, 'achieved': round(ex / dsec / 1e12, 2),
                           'peak': valu_peak, 'unit': 'TFLOP/s', 'frac': round(ex / dsec / 1e12 / valu_peak, 4),
                           'traffic': dc.get('hbm_bytes'), 'kernel_ms': round(dsec * 1e3, 4),
                           'executed_flops_per_launch': ex,
                           'algorithmic_flops_per_launch': dm['flops'],
                           'algorithmic_TFLOPs': round(dm['flops'] / dsec / 1e12, 2),
                           'algorithmic_frac': round(dm['flops'] / dsec / 1e12 / valu_peak, 4),
                           'algorithmic_bytes_per_launch': dm.get('bytes'),
                           'valu_issue_frac': drow.get('valu_issue_frac'),
                           'valu_busy_frac_4cycle': drow.get('valu_busy_frac_4cycle'), 'source': src,
                           'note': 'frac = EXECUTED flops over peak: the transforms the kernel performs (one forward per x-plane row, one '
                                   'inverse per candidate, nominal 5 L log2 L each) over the HIP-event time, against the vector peak of the precision of the plan (f32 157.3, f64 78.6 TFLOP/s); '
                                   'algorithmic_* credits the reference algorithm\'s work instead (a forward AND an inverse transform per '
                                   'candidate and row), which the shared-forward kernel does not perform; valu_issue_frac = counted VALU '
                                   'wave-instructions / (1024 SIMDs x 1 per 2 cycles x 2.4 GHz) -- a rate a SIMD reaches with 8 resident waves; '
                                   'valu_busy_frac_4cycle prices the same count at one instruction per 4 cycles, what a SIMD retires with the 2-3 '
                                   'waves this kernel\'s LDS and registers allow (profiles/r06_passA_rotation.txt section 7): on that clock the kernel '
                                   'is VALU-bound outright; a small MFMA contraction (the row-end fix) runs beside the vector pipe'}
    else:
        ach = (dc.get('hbm_bytes') or dm.get('bytes') or 0) / dsec / 1e9
        out['roofline'] = {'bound': 'hbm', 'kernel': dom, 'achieved': round(ach, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                           'frac': round(ach / HBM_PEAK_GBS, 4), 'traffic': dc.get('hbm_bytes'),
                           'algorithmic_bytes_per_launch': dm.get('bytes'), 'kernel_ms': round(dsec * 1e3, 4), 'source': src}
    # whole step: the bytes this build's kernels must move / the bytes the counters saw / the survey's model
    step_alg = sum((models[k]['bytes'] or 0) * table[k]['working_launches'] for k in table if k in models)
    step_hbm = sum(counters[k]['hbm_bytes'] * table[k]['working_launches'] for k in table if counters.get(k, {}).get('hbm_bytes'))
    out['whole_step'] = {'algorithmic_GBps': round(step_alg / res_s / 1e9, 1),
                         'counter_GBps': round(step_hbm / res_s / 1e9, 1) if step_hbm else None,
                         'counter_frac_of_hbm_peak': round(step_hbm / res_s / 1e9 / HBM_PEAK_GBS, 4) if step_hbm else None,
                         'note': 'over the resident-only step time.  (SURVEY.md 8(d)\'s byte model of the REFERENCE algorithm -- 2-D FFT '
                                 'pairs, 19 arrays per PCG iteration -- is obsolete as a yardstick: this build does not move those bytes, and '
                                 'divided by this step time the model exceeds the HBM peak; dropped from the line in round 5.)'}
    # ---- the unwrap as a group: one working PCG iteration = its five launches (stencil-fused iteration) or four; bytes of the
    #      iteration over the sum of the per-launch times (serial clock: the two components run one after the other while profiling)
    it_kernels = [k for k in ('pqdct_kernel', 'colstream_agg_kernel', 'colstream_scan_kernel', 'colstream_apply_kernel', 'rowidct_p_kernel',
                              'rowdct_fused_kernel', 'colsolve_kernel', 'colsolve_tri_kernel', 'pq_kernel', 'rowidct_pq_kernel') if k in table]
    if it_kernels and sum(iters) > 0:
        fused = 'pqdct_kernel' in table and 'colstream_agg_kernel' in table
        names = (['pqdct_kernel', 'colstream_agg_kernel', 'colstream_scan_kernel', 'colstream_apply_kernel', 'rowidct_p_kernel'] if fused
                 else [k for k in it_kernels if k not in ('pqdct_kernel',)])
        per_launch_us = {k: table[k]['total_ms'] / max(table[k]['working_launches'], 1) * 1e3 for k in names if k in table}
        it_us = sum(per_launch_us.values())
        it_bytes = (44 if fused else 48) * s // 4 * n * n     # 44 bytes per sample and iteration (f32), 48 unfused
        unwrap_ms = sum(table[k]['total_ms'] for k in it_kernels + ['phi_flush_kernel'] if k in table)
        out['roofline_unwrap'] = {'bound': 'hbm', 'group': names, 'us_per_iteration': round(it_us, 1), 'per_launch_us': {k: round(v, 1) for k, v in per_launch_us.items()},
                                  'algorithmic_bytes_per_iteration': int(it_bytes), 'achieved': round(it_bytes / (it_us * 1e-6) / 1e9, 1),
                                  'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(it_bytes / (it_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                                  'unwrap_serial_ms_both_components': round(unwrap_ms, 4),
                                  'share_of_serial_step': round(unwrap_ms / max(sum(table[k]['total_ms'] for k in table), 1e-9), 3),
                                  'note': 'one working PCG iteration as a group: its launches one after the other (HIP events) over its algorithmic '
                                          'bytes; in the timed step the two components run on two streams and overlap'}
    return out


def single_gpu(args):
    n = args.size
    knx, kny = (int(v) for v in args.kgrid.split('x')) if args.kgrid else (args.kside, args.kside)
    np_dt = np.float32 if args.dtype == 'f32' else np.float64
    if args.only_pipeline:
        print(json.dumps({'pipeline_end_to_end': pipeline_end_to_end(n, knx, kny, np_dt, args.kmax)}), flush=True)
        return
    m = measure(n, knx, kny, np_dt, args.kmax, args.steps, args.warmup, depth=args.inflight)
    P, K, sigma, kvecs, klists = m['P'], m['K'], m['sigma'], m['kvecs'], m['klists']
    out = {
        'metric': 'Mpixels/s displacement-field extraction (3 peaks, 4096^2 img) + achieved HBM GB/s',
        'value': m['value'],
        'unit': 'Mpixels/s',
        'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': m['ms_per_step'],
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': args.dtype, 'data': 'synthetic',
        'config': {'workload': '%dx%d synthetic hex moire, 3 Bragg peaks x %d k-vectors, sigma=%d, weighted DCT-PCG '
                               'unwrap kmax=%d, image resident in HBM, u downloaded to pinned host memory inside the '
                               'step (BASELINE.json configs[2])' % (n, n, K, sigma, args.kmax),
                   'image': [n, n], 'peaks': P, 'kvectors_per_peak': K, 'x_planes': m['Bx'],
                   'unwrap_iters': m['iters'], 'stopping_test': "the reference's alone (kmax or ||r|| < 1e-9 ||r0||); the library's f32 residual floor is off",
                   'd2h_of_u': 'included, overlapped with the next step (copy stream)',
                   'images_in_flight': m['depth']},
        'resident_only': {'value': m['resident_value'], 'ms_per_step': m['resident_ms'],
                          'note': 'same loop with u left in HBM (round-1 definition)'},
    }
    for key in ('early_stop', 'kernels', 'stage_ms', 'roofline', 'roofline_unwrap', 'whole_step'):
        if key in m:
            out[key] = m[key]

    if not args.no_f64 and args.dtype == 'f32':
        k64 = max(3, args.steps)     # (as many steps as the headline: the drain of the last image's 256 MB download is amortised alike)
        m64 = measure(n, knx, kny, np.float64, args.kmax, k64, max(2, args.warmup), profile=True)
        out['f64'] = {'value': m64['value'], 'unit': 'Mpixels/s', 'ms_per_step': m64['ms_per_step'],
                      'resident_only': m64['resident_value'], 'steps': k64, 'unwrap_iters': m64['iters'],
                      'note': 'the reference computes in complex128 and the mirror defaults to float64: the same step in f64, D2H of u '
                              'included, with its own per-kernel table and rooflines (VERDICT r05 item 6)'}
        for key in ('kernels', 'stage_ms', 'roofline', 'roofline_unwrap', 'whole_step'):
            if key in m64:
                out['f64'][key] = m64[key]
    if not args.no_f64 and args.dtype == 'f32' and n == 4096 and not args.kgrid:
        # BASELINE.json configs[1] (2048^2, 3 peaks x 8 k-vectors as the survey's 4 x 2 list, f32) in the same run
        # (a 2048^2 step is 1.4 ms: three times the steps of the headline, so that one slow D2H does not move the leg by 4 %)
        c2 = measure(2048, 4, 2, np.float32, args.kmax, max(30, 3 * args.steps), 3)
        out['config2'] = {'workload': '2048x2048 synthetic hex moire, 3 Bragg peaks x 8 k-vectors (4 x 2 list), f32, kmax=%d, '
                                      'D2H of u included (BASELINE.json configs[1])' % args.kmax,
                          'value': c2['value'], 'unit': 'Mpixels/s', 'ms_per_step': c2['ms_per_step'],
                          'resident_only': c2['resident_value'], 'unwrap_iters': c2['iters'],
                          'early_stop': c2.get('early_stop'), 'roofline': c2['roofline'], 'whole_step': c2['whole_step'],
                          'kernels_ms': {k: v['total_ms'] for k, v in c2['kernels'].items()}}
        out['small_image_stacks'] = small_image_stacks(kvecs, klists, sigma, args.kmax)
    if not args.no_f64:
        out['host_call'] = host_call(n, knx, kny, np_dt, args.kmax)
    if not args.no_f64 and not args.no_config5 and args.dtype == 'f32' and n == 4096 and not args.kgrid:
        try:
            out['config5_single_gpu'] = config5_single_gpu(knx, kny, np_dt, args.kmax)
        except Exception as e:      # (an extra leg must not take the headline line with it)
            out['config5_single_gpu'] = {'error': '%s: %s' % (type(e).__name__, e)}
    if not args.no_f64 and not args.no_pipeline:
        try:
            out['pipeline_end_to_end'] = pipeline_end_to_end(n, knx, kny, np_dt, args.kmax)
        except Exception as e:      # (an extra leg must not take the headline line with it)
            out['pipeline_end_to_end'] = {'error': '%s: %s' % (type(e).__name__, e)}
    if not args.no_cpu:
        out['cpu_baseline'] = cpu_baseline(kvecs, sigma, knx, kny, args.kmax, n)
    print(json.dumps(out), flush=True)


def config5_single_gpu(knx, kny, np_dt, kmax, n=16384, window=2048, reps=2):
    """BASELINE.json configs[4] on ONE GPU, end to end and resident: the 16384^2 image through the tile pipeline (81 halo
    windows of 2048^2 read in place from the resident image, interiors of the gradient fields written straight into the
    global fields, two global weighted unwraps) PLUS the Lawler-Fujita undistortion of the image with the field just
    extracted (gpa_undistort_image_dev: u never leaves HBM).  C ABI only -- no torch in this leg.  Extra key, never `value`."""
    from pygpa_amd import _lib
    from pygpa_amd import distributed as D
    from pygpa_amd.synthetic import hex_kvecs, explicit_klists
    shape = (n, n)
    P, K = 3, knx * kny
    kvecs = hex_kvecs(0.1, 7.0)
    sigma = int(np.ceil(1 / np.linalg.norm(kvecs, axis=1).min()))
    kw = np.linalg.norm(kvecs, axis=1).mean() / 2.5
    klists = np.stack(explicit_klists(kvecs, kw, knx, kny))
    halo, border = 3 * sigma, 2 * sigma
    tiles, (t0, t1), wshape = D.tile_plan(shape, None, halo, (window, window))
    rsz = np.dtype(np_dt).itemsize
    npx = n * n
    # the synthetic image, generated in bands of rows (the full-size float64 temporaries would be 2 GB each)
    t_gen = time.perf_counter()
    d_img = _lib.DeviceBuffer(npx * rsz)
    band = 1024
    y = (np.arange(n) - n // 2)[None, :].astype(np.float64)
    for r0 in range(0, n, band):
        x = (np.arange(r0, r0 + band) - n // 2)[:, None].astype(np.float64)
        ux = 0.5 * x * np.exp(-0.5 * ((x / (n / 8.0)) ** 2 + 1.2 * (y / (n / 6.0)) ** 2))
        blk = np.zeros((band, n))
        for kx, ky in kvecs:
            blk += np.cos(2 * np.pi * (kx * (x + ux) + ky * y))
        blk += np.random.default_rng([100, r0]).normal(scale=0.05, size=blk.shape)
        d_img.upload_at(np.ascontiguousarray(blk, dtype=np_dt), r0 * n * rsz)
    t_gen = time.perf_counter() - t_gen
    plan_w = _lib.Plan(wshape, P * K, np_dt, device=0)
    plan_g = _lib.Plan(shape, 1, np_dt, device=0)
    plan_g2 = _lib.Plan(shape, 1, np_dt, device=0)
    gdx, gdy = _lib.DeviceBuffer(2 * n * (n - 1) * rsz), _lib.DeviceBuffer(2 * (n - 1) * n * rsz)
    gw, d_u = _lib.DeviceBuffer(npx * rsz), _lib.DeviceBuffer(2 * npx * rsz)
    d_rec, d_uinv = _lib.DeviceBuffer(npx * rsz), _lib.DeviceBuffer(2 * npx * rsz)
    st = {}

    def one_image(record=False):
        t = time.perf_counter()
        mean = plan_w.mean_dev(d_img.ptr, npx)                     # (one scalar to the host: the only sync before the end)
        if record:
            st['mean'] = time.perf_counter() - t
            t = time.perf_counter()
        for (i, j), (w0, w1), (o0, o1), (z0, z1) in tiles:
            gi, gj = i * t0, j * t1
            plan_w.tile_gradients_dev(d_img.ptr, n, w0.start, w1.start, mean, kvecs, klists, sigma, border, (o0, o1, z0, z1),
                                      (gdx.ptr + (gi * (n - 1) + gj) * rsz, n - 1, n * (n - 1)),
                                      (gdy.ptr + (gi * n + gj) * rsz, n, (n - 1) * n), (gw.ptr + (gi * n + gj) * rsz, n))
        plan_w.sync()
        if record:
            st['tile_stage'] = time.perf_counter() - t
            t = time.perf_counter()
        # the two components on two plans (streams + workspaces) at once, as the fused driver and run_stream do
        for c, pl in enumerate((plan_g, plan_g2)):
            pl.unwrap_prediff_enqueue_dev(gdx.ptr + c * n * (n - 1) * rsz, gdy.ptr + c * (n - 1) * n * rsz, gw.ptr,
                                          d_u.ptr + c * npx * rsz, kmax=kmax)
        its = [pl.unwrap_finish() for pl in (plan_g, plan_g2)]
        if record:
            st['global_unwrap'] = time.perf_counter() - t
            t = time.perf_counter()
        plan_g.undistort_image_dev(d_img.ptr, d_u.ptr, d_rec.ptr, uinv_ptr=d_uinv.ptr, scale=-1.0)   # (extracted field = -displacement)
        plan_g.sync()
        if record:
            st['lawler_fujita'] = time.perf_counter() - t
        return its

    one_image()
    t0_ = time.perf_counter()
    for _ in range(reps):
        its = one_image()
    dt = (time.perf_counter() - t0_) / reps
    one_image(record=True)
    plan_g.set_profiling(True)
    plan_g.undistort_image_dev(d_img.ptr, d_u.ptr, d_rec.ptr, uinv_ptr=d_uinv.ptr, scale=-1.0)
    prof = plan_g.last_kernel_profile()
    plan_g.set_profiling(False)
    # compulsory HBM bytes of the Lawler-Fujita kernels per launch: pad / FIR passes read + write one padded
    # field; the fixed point reads the two coefficient fields once and writes u_inv; the resampling reads coefficients +
    # u_inv and writes the image.  The 36 rounds x 32 taps per pixel of the fixed point are CACHE traffic (L1 / L2).
    mpx = (n + 24) * (n + 24)
    comp = {'pad_edge_kernel': 2 * mpx, 'fir_rows_kernel': 2 * mpx, 'fir_cols_kernel': 2 * mpx, 'invert_kernel': 2 * mpx + 2 * npx,
            'warp_constant_kernel': 4 * npx}
    lf = {}
    for k, (calls, ms) in prof.items():
        gb = comp.get(k, 0) * rsz / 1e9
        lf[k] = {'launches': calls, 'total_ms': round(ms, 3), 'compulsory_GB_per_launch': round(gb, 3),
                 'frac_of_hbm_peak': round(gb * calls / (ms * 1e-3) / HBM_PEAK_GBS, 3) if ms else None}
    if 'invert_kernel' in lf:
        lf['invert_kernel']['bound'] = ('instruction issue, not bytes: the reference runs 36 rounds of 2 x 16-tap cubic interpolations per pixel; a '
                                        'wavefront leaves once every pixel is at a bitwise fixed point or in a cycle of two (mean ~5 rounds, same '
                                        'bits), the first rounds gather from L1 / L2 and the later ones from a 48 x 48 LDS window per 16 x 16 tile: '
                                        '~1400 vector instructions per wavefront (tools/lf_pmc.sh); compulsory HBM bytes are a few percent of the time')
    for b in (d_img, gdx, gdy, gw, d_u, d_rec, d_uinv):
        b.free()
    plan_w.close()
    plan_g.close()
    plan_g2.close()
    return {'workload': '%dx%d synthetic hex moire, 3 x %d k-vectors, %s: %d halo windows of %d^2 (halo %d) read in place -> gradient '
                        'interiors -> two global weighted unwraps kmax=%d -> Lawler-Fujita undistortion with the extracted field '
                        '(36 fixed-point rounds + cubic resampling); one GPU, everything resident (BASELINE.json configs[4] end to end)'
                        % (n, n, K, np.dtype(np_dt).name, len(tiles), window, halo, kmax),
            'value': round(npx / dt / 1e6, 1), 'unit': 'Mpixels/s', 'ms_per_image': round(dt * 1e3, 2), 'unwrap_iters': list(its),
            'stage_ms': {k: round(v * 1e3, 2) for k, v in st.items()},
            'extraction_only_Mpix_s': round(npx / max(dt - st.get('lawler_fujita', 0.0), 1e-9) / 1e6, 1),
            'lawler_fujita_kernels': lf, 'image_generation_s': round(t_gen, 1)}


# compulsory HBM words (reals of the plan dtype) per pixel and launch of the kernels either side of the hot path that are pure
# streams (P = 3 peaks): what the fraction of the HBM peak in `pipeline_end_to_end` is measured against
PIPE_WORDS = {
    'dft_rows_r2c_kernel': 2.0,      # image in, bins 0 ... n/2 of every row out (half of u_hat: |P^| is symmetric)
    'dft_cols_kernel': 2.0,          # that half in place: in + out
    'per_absshift_kernel': 2.0,      # half of u_hat in, |fftshift(P^)| out
    'gauss2d_small_kernel': 2.0,     # in, out
    'gauss_fft_cols_kernel': 2.0,    # in, out (overlap-save re-reads 2R rows per segment from L2)
    'gauss_fft_rows_kernel': 3.0,    # in, minuend, out
    'minmax_kernels': 1.0, 'localmax_kernel': 1.0,
    'lockin_abs_kernel': 9.0,        # 3 complex lock-ins in, 3 weights out
    'jacobian_kernel': 13.0,         # 3 gradient pairs + 3 weights in, J (2 x 2) out
    'props_kernel': 8.0,             # J in, 4 properties out
    'fir_rows_kernel': 2.0, 'fir_cols_kernel': 2.0, 'warp_constant_kernel': 4.0,
}


def pipeline_end_to_end(n, knx, kny, np_dt, kmax, reps=3):
    """What a user of the reference runs around the hot path, resident on one GPU at the headline's size:
    extract_primary_ks (geometric_phase_analysis.py:397-505) -> extract_displacement_field with the k-vectors just found
    (:907-932) -> undistort_image (:935-974) -> wfr2_grad_opt per peak (:763-813) + phasegradient2J + props_from_Jac
    (property_extract.py:69-101, :137-178).  Per stage: wall clock around the calls with one stream sync; per kernel: HIP
    events on the launch stream (gpa_set_profiling), and for the pure streams the fraction of the HBM peak on their compulsory
    bytes.  Extra key, never `value`."""
    from pygpa_amd import _lib
    import pygpa_amd.geometric_phase_analysis as GPA
    from pygpa_amd.synthetic import hex_kvecs, explicit_klists, gaussian_bump_displacement, hex_moire
    P, K = 3, knx * kny
    true_ks = hex_kvecs(0.1, 7.0)
    img = hex_moire((n, n), true_ks, gaussian_bump_displacement((n, n)), noise=0.1, seed=100, dtype=np.float64)
    img = (img - img.mean()).astype(np_dt)          # the callers of wfr2_grad_opt subtract the mean (:919)
    s = np.dtype(np_dt).itemsize
    npx = n * n
    plan = _lib.Plan((n, n), P * K, np_dt, device=0)
    bufs = {k: _lib.DeviceBuffer(v * npx * s) for k, v in dict(img=1, u=2, rec=1, uinv=2, lock=2 * P, grad=2 * P, w=P, J=4, props=4).items()}
    bufs['img'].upload(img)
    st = {}

    def stage_ks(profile=None):
        if profile is not None:      # every evaluation of the relaxation loop is one library call with its own kernel profile
            def recording(fn):
                def call(*a, **k):
                    r = fn(*a, **k)
                    for name, (c, ms) in plan.last_kernel_profile().items():
                        c0, t0 = profile.get(name, (0, 0.0))
                        profile[name] = (c0 + c, t0 + ms)
                    profile['(library calls)'] = (profile.get('(library calls)', (0, 0.0))[0] + 1, 0.0)
                    return r
                return call
            plan.find_peaks_dev, plan.find_peaks_again = recording(plan.find_peaks_dev), recording(plan.find_peaks_again)
        try:
            ks, _ = GPA.extract_primary_ks_dev(plan, bufs['img'].ptr, pix_norm_range=(2, 0.2 * n))
        finally:
            if profile is not None:
                del plan.find_peaks_dev, plan.find_peaks_again
        return ks

    ks = stage_ks()
    sigma = int(np.ceil(1 / np.linalg.norm(ks, axis=1).min()))
    kw = np.linalg.norm(ks, axis=1).mean() / 2.5
    klists = np.stack(explicit_klists(ks, kw, knx, kny))

    def stage_extract():
        plan.extract_displacement_field_dev(bufs['img'].ptr, ks, klists, sigma, 2 * sigma, kmax, bufs['u'].ptr)

    def stage_undistort():
        # (the extracted field is MINUS the displacement, tests/test_geometric_phase_analysis.py:63: undistort with -u)
        plan.undistort_image_dev(bufs['img'].ptr, bufs['u'].ptr, bufs['rec'].ptr, uinv_ptr=bufs['uinv'].ptr, scale=-1.0)

    def stage_props(profile=None):
        def note():
            if profile is not None:
                for name, (c, ms) in plan.last_kernel_profile().items():
                    c0, t0 = profile.get(name, (0, 0.0))
                    profile[name] = (c0 + c, t0 + ms)
        for p_ in range(P):
            plan.sweep_grad_dev(bufs['img'].ptr, ks[p_], klists[p_], sigma, bufs['lock'].ptr + p_ * 2 * npx * s,
                                bufs['grad'].ptr + p_ * 2 * npx * s)
            note()
        plan.lockin_weights_dev(bufs['lock'].ptr, P, bufs['w'].ptr)
        note()
        plan.phasegradient2J_dev(ks, bufs['grad'].ptr, bufs['w'].ptr, 1.0, bufs['J'].ptr)
        note()
        plan.timer_start()
        plan.props_from_jac_dev(bufs['J'].ptr, bufs['props'].ptr, add_identity=True)
        ms = plan.timer_stop()
        if profile is not None:
            profile['props_kernel'] = (1, ms)

    stages = [('extract_primary_ks', stage_ks), ('extract_displacement_field', stage_extract), ('undistort_image', stage_undistort),
              ('wfr2_grad_opt_x3+phasegradient2J+props_from_Jac', stage_props)]
    for _, f in stages:
        f()
    plan.sync()
    t_all = time.perf_counter()
    for _ in range(reps):
        for name, f in stages:
            t = time.perf_counter()
            f()
            plan.sync()
            st[name] = st.get(name, 0.0) + (time.perf_counter() - t) / reps
    t_all = (time.perf_counter() - t_all) / reps
    # per-kernel HIP-event times of one profiled pass
    plan.set_profiling(True)
    kern = {}
    for name, f in stages:
        prof = {}
        if name.startswith('wfr2') or name == 'extract_primary_ks':
            f(prof)
        else:
            f()
            plan.sync()
            prof = plan.last_kernel_profile()
        kern[name] = prof
    plan.set_profiling(False)
    table = {}
    for stage, prof in kern.items():
        rows = {}
        for k, (calls, ms) in prof.items():
            row = {'launches': calls, 'total_ms': round(ms, 4)}
            if k in PIPE_WORDS and ms > 0:
                gb = PIPE_WORDS[k] * s * npx * calls / 1e9
                row['compulsory_GB'] = round(gb, 3)
                row['frac_of_hbm_peak'] = round(gb / (ms * 1e-3) / HBM_PEAK_GBS, 3)
            rows[k] = row
        table[stage] = rows
    err = float(np.abs(np.minimum(np.linalg.norm(ks[:, None] - true_ks[None], axis=2), np.linalg.norm(ks[:, None] + true_ks[None], axis=2)).min(axis=1)).max())
    for b in bufs.values():
        b.free()
    plan.close()
    return {'workload': '%dx%d synthetic hex moire resident in HBM, %s: extract_primary_ks (DoG, threshold relaxation on the host) -> '
                        'extract_displacement_field (3 x %d k-vectors around the k-vectors FOUND, kmax=%d) -> undistort_image '
                        '(Lawler-Fujita with the extracted field) -> wfr2_grad_opt per peak + phasegradient2J + props_from_Jac; '
                        'nothing but the candidate lists leaves the device' % (n, n, np.dtype(np_dt).name, K, kmax),
            'value': round(npx / t_all / 1e6, 1), 'unit': 'Mpixels/s', 'ms_per_image': round(t_all * 1e3, 3),
            'stage_ms': {k: round(v * 1e3, 3) for k, v in st.items()},
            'found_kvectors_max_error_cycles_per_px': err, 'reference_bar': 1.5 / n,
            'kernels': table,
            'note': 'stage_ms: host wall clock with one stream synchronisation per stage; kernels: HIP events of one profiled pass of '
                    'each stage (the last evaluation of extract_primary_ks\' relaxation loop); frac_of_hbm_peak on the compulsory '
                    'bytes of PIPE_WORDS, for the pure streams only'}


def host_call(n, knx, kny, np_dt, kmax, reps=5):
    """SURVEY.md 8(d) "H2D reported separately": the call a user of the reference's NumPy-in / NumPy-out signature makes
    (geometric_phase_analysis.py:907-932) -- Plan.extract_displacement_field on HOST arrays, H2D of the image and D2H of u
    inside the timed call -- with pageable arrays and with page-locked ones.  Never `value`."""
    from pygpa_amd import _lib
    from pygpa_amd.synthetic import hex_kvecs, explicit_klists, gaussian_bump_displacement, hex_moire
    kvecs = hex_kvecs(0.1, 7.0)
    sigma = int(np.ceil(1 / np.linalg.norm(kvecs, axis=1).min()))
    kw = np.linalg.norm(kvecs, axis=1).mean() / 2.5
    klists = np.stack(explicit_klists(kvecs, kw, knx, kny))
    img = hex_moire((n, n), kvecs, gaussian_bump_displacement((n, n)), noise=0.1, seed=100, dtype=np_dt)
    plan = _lib.Plan((n, n), 3 * knx * kny, np_dt, device=0)
    out = {}
    for kind in ('pageable', 'pinned'):
        if kind == 'pinned':
            src, dst = _lib.pinned_empty((n, n), np_dt), _lib.pinned_empty((2, n, n), np_dt)
            src[...] = img
        else:
            src, dst = img, np.empty((2, n, n), np_dt)
        plan.extract_displacement_field(src, kvecs, klists, sigma, 2 * sigma, kmax=kmax, out=dst)
        t0 = time.perf_counter()
        for _ in range(reps):
            plan.extract_displacement_field(src, kvecs, klists, sigma, 2 * sigma, kmax=kmax, out=dst)
        dt = (time.perf_counter() - t0) / reps
        out[kind] = {'value': round(n * n / dt / 1e6, 1), 'ms_per_call': round(dt * 1e3, 3)}
    plan.close()
    out['unit'] = 'Mpixels/s'
    out['note'] = ('one synchronous host-array call per image: H2D of the image + the step + D2H of u inside the timed call, nothing '
                   'overlapped between calls (PCIe-inclusive rate; `value` has the image resident in HBM)')
    return out


def small_image_stacks(kvecs, klists, sigma, kmax, sizes=(512, 1024), stack=16, reps=6):
    """extra key of the default run (a fraction of a second): the sizes the reference's users run most, one image
    per call against a stack of `stack` frames in ONE call (gpa_extract_displacement_field_batch_dev: every kernel
    takes an image index from its grid), images and u resident in HBM, f32, same 3 x 16 workload"""
    from pygpa_amd import _lib
    from pygpa_amd.synthetic import gaussian_bump_displacement, hex_moire
    res = {'stack': stack, 'note': 'Mpixels/s, u left in HBM; single = one image per driver call, stacked = %d frames per call' % stack}
    for n in sizes:
        img = hex_moire((n, n), kvecs, gaussian_bump_displacement((n, n)), noise=0.1, seed=1, dtype=np.float32)
        plan = _lib.Plan((n, n), klists.shape[0] * klists.shape[1], np.float32, device=0)
        frames = np.stack([img] * stack)
        d, u = _lib.DeviceBuffer(frames.nbytes), _lib.DeviceBuffer(2 * frames.nbytes)
        d.upload(frames)
        t = {}
        for mode in ('single', 'stacked'):
            def run(k):
                for _ in range(k):
                    if mode == 'single':
                        plan.extract_displacement_field_async(d.ptr, kvecs, klists, sigma, 2 * sigma, kmax, u.ptr)
                    else:
                        plan.extract_displacement_field_batch_dev(d.ptr, stack, kvecs, klists, sigma, 2 * sigma, kmax, u.ptr,
                                                                  want_iters=False)
                plan.sync()
            run(2)
            k = reps * (stack if mode == 'single' else 1)
            t0 = time.perf_counter()
            run(k)
            t[mode] = (time.perf_counter() - t0) / (k if mode == 'single' else k * stack)
        res['%dx%d' % (n, n)] = {'single': round(n * n / t['single'] / 1e6, 1), 'stacked': round(n * n / t['stacked'] / 1e6, 1)}
        d.free()
        u.free()
        plan.close()
    return res


# -------------------------------------------------------------------------------------------------
# N > 1: tile pipeline over the ranks
# -------------------------------------------------------------------------------------------------
def weak_shape(world, n):
    """image of world * n^2 pixels, as square as powers of two allow (N = 4: 2n x 2n = BASELINE configs[3]); N = 8 runs
    BASELINE configs[4]'s 4n x 4n = 16384^2 image (2 n^2 pixels per GPU: the throughput in Mpixels/s stays comparable)"""
    if world == 8:
        return (4 * n, 4 * n)
    a = 1
    while a * a * 2 <= world:
        a *= 2
    b = world // a
    if a * b != world:
        a, b = world, 1
    return (a * n, b * n) if a >= b else (b * n, a * n)


def multi_gpu(args, world, rank, local_rank):
    import torch
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dev_index = 0 if args.share_device else local_rank
    torch.cuda.set_device(dev_index)
    if args.backend == 'nccl':
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', dev_index))
    else:
        dist.init_process_group(args.backend, rank=rank, world_size=world)
    # the line says n_gpus = N only if the backend's process group really has N ranks
    if dist.get_world_size() != args.gpus or dist.get_world_size() != world:
        raise SystemExit('bench.py --gpus %d: the %s process group reports %d ranks (WORLD_SIZE=%d)'
                         % (args.gpus, args.backend, dist.get_world_size(), world))
    from pygpa_amd import distributed as D
    from pygpa_amd import _lib
    from pygpa_amd.synthetic import hex_kvecs, explicit_klists
    n = args.size
    knx, kny = (int(v) for v in args.kgrid.split('x')) if args.kgrid else (args.kside, args.kside)
    P, K = 3, knx * kny
    np_dt = np.float32 if args.dtype == 'f32' else np.float64
    kvecs = hex_kvecs(0.1, 7.0)
    sigma = int(np.ceil(1 / np.linalg.norm(kvecs, axis=1).min()))
    kw = np.linalg.norm(kvecs, axis=1).mean() / 2.5
    klists = np.stack(explicit_klists(kvecs, kw, knx, kny))
    shape = weak_shape(world, n)
    halo = 3 * sigma
    W = min(args.window, min(shape))
    pipe = D.TiledPipeline(shape, kvecs, klists, sigma, halo, kmax=args.kmax, dtype=np_dt, device=dev_index,
                           window=(W, W))

    # synthetic image: every rank generates only the pixels of its own windows (global coordinates)
    def window_fn(w0, w1):
        x = (np.arange(w0.start, w0.stop) - shape[0] // 2)[:, None].astype(np.float64)
        y = (np.arange(w1.start, w1.stop) - shape[1] // 2)[None, :].astype(np.float64)
        ux = 0.5 * x * np.exp(-0.5 * ((x / (shape[0] / 8.0)) ** 2 + 1.2 * (y / (shape[1] / 6.0)) ** 2))
        img = np.zeros((w0.stop - w0.start, w1.stop - w1.start))
        for kx, ky in kvecs:
            img += np.cos(2 * np.pi * (kx * (x + ux) + ky * y))
        rng = np.random.default_rng([100, w0.start, w1.start])
        return (img + rng.normal(scale=0.05, size=img.shape)).astype(np_dt)

    pipe.load(window_fn=window_fn)
    dev = torch.device('cuda', dev_index)

    def fence():
        torch.cuda.synchronize(dev)
        dist.barrier()
        torch.cuda.synchronize(dev)

    # one step = one image through the tile pipeline.  Default: the image-pipelined schedule (run_stream: the global
    # unwrap of image i on the rotating owners (2 i + c) % N beside the sweeps of image i + 1); --schedule step = the
    # unpipelined step() (unwrap on ranks 0 / 1 while the others wait, u broadcast to every rank)
    stream = args.schedule == 'stream'
    if stream:
        # (untimed) at least one image per pair of unwrap owners, so that every point-to-point route has been used once
        pipe.run_stream([None] * max(args.warmup, (world + 1) // 2, 1))
    else:
        for _ in range(args.warmup):
            pipe.step()
    fence()
    t0 = time.perf_counter()
    if stream:
        its = pipe.run_stream([None] * args.steps)
    else:
        for _ in range(args.steps):
            pipe.step()
    fence()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64)
    if args.backend == 'nccl':
        t = t.to(dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    # ---- diagnostics for the scaling line (after the timed region): what the backend says the world is, and per-rank
    #      stage times -- host times of the timed stream (enqueueing + waits) and DEVICE-inclusive times of one unpipelined
    #      step with the device drained after every stage
    #      (the timed value is complete at this point: a diagnostic that fails on every rank alike is reported, not fatal)
    dev_ms, diag_err = {}, None
    try:
        pipe.step(timings=dev_ms)
    except Exception as e:
        diag_err = 'step(timings): %s: %s' % (type(e).__name__, e)
    lf_ms = None
    if not args.no_lf and diag_err is None:
        try:
            torch.cuda.synchronize(dev)
            t_lf = time.perf_counter()
            full = torch.zeros(shape, dtype=torch.float32 if np_dt is np.float32 else torch.float64, device=dev)
            pipe.undistort(full)          # (the rounds depend on u, not on the image: a zero image times the same kernels)
            torch.cuda.synchronize(dev)
            lf_ms = (time.perf_counter() - t_lf) * 1e3
        except Exception as e:
            diag_err = 'undistort: %s: %s' % (type(e).__name__, e)
    mine = {'rank': rank, 'tiles': len(pipe.mine), 'stream_host_ms_per_image': {k: round(v / max(args.steps, 1) * 1e3, 3) for k, v in pipe.stage_s.items()} if stream else None,
            'step_device_ms': {k: round(v, 3) for k, v in dev_ms.items()}, 'undistort_tile_sharded_ms': None if lf_ms is None else round(lf_ms, 2),
            'diagnostic_error': diag_err}
    per_rank = [None] * world
    dist.all_gather_object(per_rank, mine)
    if rank == 0:
        npx = shape[0] * shape[1]
        cfg = {4: 'BASELINE.json configs[3]', 8: 'BASELINE.json configs[4]: the 16384^2 image, WITHOUT the Lawler-Fujita '
                  'undistortion (a per-image host call on the rank that ends up with u, not part of this step)'}.get(world, 'configs[3] pipeline')
        if stream:
            unw = [it for it in its if it[0] is not None or it[1] is not None]
            sched = ('image-pipelined: gather of 3 of 5 gradient fields to each of the two unwrap owners (2 i + c) %% N of image i, '
                     'global weighted unwrap kmax=%d there beside the sweeps of image i + 1, component 1 handed to the owner of '
                     'component 0' % args.kmax)
            coll = 'all_reduce(mean scalar), 2 x gather(3 gradient fields of every tile -> one owner), send/recv(u component)'
            stages = {k: round(v / args.steps * 1e3, 3) for k, v in pipe.stage_s.items()}
            iters_rep = [list(it) for it in unw[:2]]
        else:
            sched = 'all_gather of the gradient tiles, global weighted unwrap kmax=%d on rank c %% N, broadcast of u' % args.kmax
            coll = 'all_reduce(mean scalar), all_gather(gradient tiles), 2 x broadcast(u component)'
            stages, iters_rep = None, list(pipe.iters)
        out = {
            'metric': 'Mpixels/s displacement-field extraction (3 peaks, 4096^2 img) + achieved HBM GB/s',
            'value': round(npx * args.steps / dt / 1e6, 2), 'unit': 'Mpixels/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(dt / args.steps * 1e3, 4),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': args.dtype, 'data': 'synthetic',
            #  N = 8 runs BASELINE configs[4]'s 16384^2 image: 2 x the pixels per GPU of N = 1, 2, 4
            'weak_scaling_pixels_per_gpu': shape[0] * shape[1] // world,
            'weak_scaling_note': None if world != 8 else 'N = 8 is configs[4] (16384^2): 2 x 4096^2 pixels per GPU, twice the per-GPU '
                                 'work of the N = 1 / 2 / 4 points -- compare Mpixels/s, not a strict weak-scaling efficiency',
            'config': {'workload': '%dx%d synthetic hex moire (%.3g x %d^2 pixels per GPU) tiled into %d halo windows of %d^2 dealt '
                                   'over %d ranks, 3 Bragg peaks x %d k-vectors, sigma=%d; %s (%s); windows resident in HBM'
                                   % (shape[0], shape[1], shape[0] * shape[1] / float(world * n * n), n, len(pipe.tiles), W, world, K,
                                      sigma, sched, cfg),
                       'image': list(shape), 'tiles': len(pipe.tiles), 'window': [W, W], 'halo': halo,
                       'tile_interior': list(pipe.tshape), 'peaks': P, 'kvectors_per_peak': K,
                       'pixels_per_gpu': shape[0] * shape[1] // world, 'schedule': args.schedule,
                       'unwrap_iters': iters_rep, 'backend': args.backend, 'collectives': coll,
                       'ranks_in_process_group': dist.get_world_size(), 'parallelism': 'tiles over %d ranks' % dist.get_world_size(),
                       'stage_ms_per_image_rank0': stages},
            'ranks': {'world_size_reported_by_backend': dist.get_world_size(), 'backend': dist.get_backend(),
                      'rccl_version': list(torch.cuda.nccl.version()) if args.backend == 'nccl' else None,
                      'devices_visible': torch.cuda.device_count(), 'per_rank': per_rank,
                      'note': 'stream_host_ms = host time of the timed image stream per stage (enqueueing and waits); step_device_ms = one '
                              'unpipelined step() after the timed region with the device drained after every stage; '
                              'undistort_tile_sharded_ms = TiledPipeline.undistort (Lawler-Fujita of the stitched field, every rank its own '
                              'tiles, one all_reduce), first call, after the timed region -- NOT part of `value`'},
        }
        # roofline of the tile stage's dominant kernel (pass B on one window), measured on this rank AFTER the timed
        # region on a whole-image call of the window's shape: the same kernel instantiation on the same rows
        try:
            torch.cuda.synchronize(dev)
            m = measure(W, knx, kny, np_dt, args.kmax, 5, 1, profile=True)
            rf = dict(m['roofline'])
            rf['note'] = ('dominant kernel of the tile stage, from a %d^2 whole-image call on rank 0 after the timed region '
                          '(same kernel, same window shape); ' % W) + str(rf.get('note', ''))
            out['roofline'] = rf
        except Exception as e:   # the scaling line must not die of its diagnostics
            out['roofline'] = None
            out['roofline_error'] = '%s: %s' % (type(e).__name__, e)
        print(json.dumps(out), flush=True)
    pipe.close()
    dist.barrier()
    dist.destroy_process_group()


def spawn(args):
    """--gpus N without a launcher: start N fresh rank processes BEFORE this process touches the GPU,
    relay rank 0's JSON line and exit with the children's return code"""
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    r = subprocess.run(cmd, env=env)
    sys.exit(r.returncode)


def main():
    args = parse_args()
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        spawn(args)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world == 1:
        single_gpu(args)
    else:
        multi_gpu(args, world, int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0')))


if __name__ == '__main__':
    main()
