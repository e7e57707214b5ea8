#!/usr/bin/env python3
"""Single-GPU stage times of the tile pipeline (inputs of the multi-GPU scaling model in DESIGN.md 5):
seconds per 2048^2 halo window of the tile stage (sweep + least squares) and per component of the global weighted
unwrap at 4096^2 ... 16384^2, f32, 3 x 16 candidates, kmax 10.

    python tools/stage_times.py [--sizes 4096,8192,16384] [--window 2048]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--sizes', default='4096,8192,16384')
    ap.add_argument('--window', type=int, default=2048)
    args = ap.parse_args()
    import torch
    from pygpa_amd import distributed as D
    from pygpa_amd import _lib
    _lib.set_option('F32_EPS_FLOOR', '0')   # the reference's stopping test alone (kmax iterations here), as in bench.py
    from pygpa_amd.synthetic import hex_kvecs, explicit_klists
    kvecs = hex_kvecs(0.1, 7.0)
    sigma = 10
    kw = np.linalg.norm(kvecs, axis=1).mean() / 2.5
    klists = np.stack(explicit_klists(kvecs, kw, 4, 4))
    out = {}
    for n in (int(v) for v in args.sizes.split(',')):
        shape = (n, n)
        W = min(args.window, n)
        pipe = D.TiledPipeline(shape, kvecs, klists, sigma, 3 * sigma, kmax=10, dtype=np.float32, device=0, window=(W, W))

        def window_fn(w0, w1):
            x = (np.arange(w0.start, w0.stop) - n // 2)[:, None].astype(np.float64)
            y = (np.arange(w1.start, w1.stop) - n // 2)[None, :].astype(np.float64)
            img = np.zeros((w0.stop - w0.start, w1.stop - w1.start))
            for kx, ky in kvecs:
                img += np.cos(2 * np.pi * (kx * (x + 0.3 * x * np.exp(-0.5 * ((x / (n / 8.0)) ** 2 + (y / (n / 6.0)) ** 2))) + ky * y))
            return img.astype(np.float32)

        pipe.load(window_fn=window_fn)
        pipe.run_stream([None, None])
        reps = 4
        t0 = time.perf_counter()
        pipe.run_stream([None] * reps)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        st = {k: v / reps for k, v in pipe.stage_s.items()}
        # the unwrap alone: both components one after the other through step()'s path
        pipe.step()
        t0 = time.perf_counter()
        pipe.step()
        torch.cuda.synchronize()
        t_step = time.perf_counter() - t0
        # the stages on their own (nothing else on the GPU): what the scaling model of DESIGN.md 5 is built from
        def tile_stage():
            pipe._tile_stage()
            pipe.be.tiles_to_torch(True)
        tile_stage()
        t0 = time.perf_counter(); tile_stage(); t_tiles = time.perf_counter() - t0
        def unwrap(c):
            pipe.be.unwrap_start(c, pipe.gdx[c], pipe.gdy[c], pipe.gw, pipe.u[c], pipe.kmax)
            return pipe.be.unwrap_wait(c)
        unwrap(0)
        t0 = time.perf_counter(); it0 = unwrap(0); t_unw = time.perf_counter() - t0
        # the pipeline's own data-path kernels on an otherwise idle GPU: interior sums + mean, stitch of one component
        def timed(fn, reps=3):
            fn(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            pipe.be.sync_tiles_all()
            return (time.perf_counter() - t0) / reps
        pipe.be.sync_tiles_all = lambda: (pipe.be.plan_w.sync(), [pl.sync() for pl in pipe.be.plan_c if pl is not None], torch.cuda.synchronize())
        t_mean = timed(pipe.image_mean)
        plane = pipe.tshape[0] * pipe.tshape[1]
        t_stitch = timed(lambda: pipe.be.stitch(0, pipe.local[0], 3 * plane, pipe.table_stream, len(pipe.tiles), pipe.tshape[0], pipe.tshape[1],
                                                pipe.gdx[0], pipe.gdy[0], pipe.gw))
        iso = {'mean_s_alone': round(t_mean, 6), 'stitch_s_one_component_alone': round(t_stitch, 6),
               'tile_stage_s_all_windows': round(t_tiles, 5), 'tile_stage_s_per_window_alone': round(t_tiles / len(pipe.tiles), 6),
               'unwrap_s_one_component_alone': round(t_unw, 5), 'unwrap_iters': it0}
        print(n, 'alone:', json.dumps(iso), flush=True)
        out['%d' % n] = {'alone': iso,'windows': len(pipe.tiles), 'window': W, 'stream_s_per_image': round(dt, 5), 'step_s_per_image': round(t_step, 5),
                         'tile_stage_s_per_window': round(st['tiles'] / len(pipe.tiles), 6),
                         'stage_s': {k: round(v, 5) for k, v in st.items()}, 'iters': list(pipe.iters)}
        print(n, json.dumps(out['%d' % n]), flush=True)
        pipe.close()
        del pipe
        torch.cuda.empty_cache()
    json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'stage_times.json'), 'w'), indent=1)


if __name__ == '__main__':
    main()
