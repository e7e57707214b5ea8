#!/usr/bin/env python3
"""Predicted per-image time and Mpixels/s of the tile pipeline on 1 / 2 / 4 / 8 MI355X from MEASURED single-GPU stage
times (profiles/r06_stage_times.json, tools/stage_times.py) and a per-link xGMI model -- the curve a SCALE run of
`bench.py --gpus N` is to be compared with (no multi-GPU box was available to this build).

Model, per image in the steady state of TiledPipeline.run_stream (every rank carries the same load over a rotation):
    T = T_mean / N + ceil(W / N) t_window + (2 / N) (T_unwrap(image) + T_stitch) + T_gather + T_handover / (N / 2) + n_coll T_lat
    T_mean     = interior sums of a rank's windows (one launch) + the set-mean launch; N > 1: + one all_reduce of a double
    T_stitch   = one launch per component on its owner (3 fields of the whole image)
    T_gather   = 3 fields x 4 B x pixels / N over ONE link per (source, owner) pair        (N > 1)
    T_handover = 4 B x pixels over one link, paid by one pair per image                     (N > 1)
    n_coll     = collectives a rank takes part in per image: all_reduce + 2 gathers + (2 / N) send / recv, T_lat = 30 us each
                 (an ASSUMED launch + rendezvous latency of an RCCL operation: no multi-GPU box was available to measure it)
and of step() (all_gather of 5 fields to every rank, unwrap on ranks 0 / 1, two broadcasts):
    T = ceil(W / N) t_window + T_unwrap + 5 fields x 4 B x pixels / N per link + 2 x 4 B x pixels per link
LINK = one direction of one xGMI link (the task statement's ~153 GB/s per link counts both directions)."""
import json
import math
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINK = 153e9 / 2
st = json.load(open(os.path.join(ROOT, 'profiles', 'r06_stage_times.json')))
T_LAT = 30e-6
shapes = {1: 4096, 2: None, 4: 8192, 8: 16384}
tw = st['4096']['alone']['tile_stage_s_per_window_alone']
print('| GPUs | image | windows | mean | tile stage | unwrap + stitch share | gather | hand-over | collective latencies | T per image (stream) | Mpix/s (stream) | T (step) | Mpix/s (step) |')
print('|---|---|---|---|---|---|---|---|---|---|---|---|---|')
def interp(key, px):
    """a per-image quantity at a pixel count between the measured squares (linear in pixels; the 8192 x 4096 image of N = 2)"""
    a, b = st['4096']['alone'][key], st['8192']['alone'][key]
    f = (px - 4096 ** 2) / float(8192 ** 2 - 4096 ** 2)
    return a + f * (b - a)


for N, n in shapes.items():
    if n is None:
        px, W, name = 8192 * 4096, 15, '8192x4096'
        unw, stitch, mean = (interp(k, px) for k in ('unwrap_s_one_component_alone', 'stitch_s_one_component_alone', 'mean_s_alone'))
        unw *= 1.1      # (8192-point rows on half of the rows: nearer the 8192^2 per-pixel rate than the mean of the two)
    else:
        px, W, name = n * n, st[str(n)]['windows'], '%d^2' % n
        al = st[str(n)]['alone']
        unw, stitch, mean = al['unwrap_s_one_component_alone'], al['stitch_s_one_component_alone'], al['mean_s_alone']
    tile = math.ceil(W / N) * tw
    mean_n = mean / N
    if N == 1:
        gather = hand = lat = 0.0
        t_stream = mean_n + tile + 2 * (unw + stitch)
        t_step = t_stream
    else:
        gather = 3 * 4 * px / N / LINK
        hand = 4 * px / LINK
        lat = (1 + 2 + 2.0 / N) * T_LAT
        t_stream = mean_n + tile + 2 * (unw + stitch) / N + gather + hand / (N / 2) + lat
        t_step = mean_n + tile + unw + stitch + 5 * 4 * px / N / LINK + 2 * 4 * px / LINK + 4 * T_LAT
    print('| %d | %s | %d | %.2f ms | %.1f ms | %.1f ms | %.1f ms | %.1f ms | %.2f ms | **%.1f ms** | **%.0f** | %.1f ms | %.0f |' % (
        N, name, W, mean_n * 1e3, tile * 1e3, 2 * (unw + stitch) / N * 1e3, gather * 1e3, hand / max(N / 2, 1) * 1e3 if N > 1 else 0,
        lat * 1e3, t_stream * 1e3, px / t_stream / 1e6, t_step * 1e3, px / t_step / 1e6))
