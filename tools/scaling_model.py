#!/usr/bin/env python3
"""Predicted per-image time and Mpixels/s of the tile pipeline on 1 / 2 / 4 / 8 MI355X from MEASURED single-GPU stage
times (profiles/r03_stage_times.json, tools/stage_times.py) and a per-link xGMI model -- the curve a SCALE run of
`bench.py --gpus N` is to be compared with (no multi-GPU box was available to this build).

Model, per image in the steady state of TiledPipeline.run_stream (every rank carries the same load over a rotation):
    T = ceil(W / N) t_window + (2 / N) T_unwrap(image) + T_gather + T_handover / (N / 2)
    T_gather   = 3 fields x 4 B x pixels / N over ONE link per (source, owner) pair        (N > 1)
    T_handover = 4 B x pixels over one link, paid by one pair per image                     (N > 1)
and of step() (all_gather of 5 fields to every rank, unwrap on ranks 0 / 1, two broadcasts):
    T = ceil(W / N) t_window + T_unwrap + 5 fields x 4 B x pixels / N per link + 2 x 4 B x pixels per link
LINK = one direction of one xGMI link (the task statement's ~153 GB/s per link counts both directions)."""
import json
import math
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINK = 153e9 / 2
st = json.load(open(os.path.join(ROOT, 'profiles', 'r03_stage_times.json')))
shapes = {1: 4096, 2: None, 4: 8192, 8: 16384}
tw = st['4096']['alone']['tile_stage_s_per_window_alone']
print('| GPUs | image | windows | tile stage | unwrap share | gather | hand-over | T per image (stream) | Mpix/s (stream) | T (step) | Mpix/s (step) |')
print('|---|---|---|---|---|---|---|---|---|---|---|')
base = None
for N, n in shapes.items():
    if n is None:
        # 8192 x 4096: windows and unwrap time interpolated by pixels between the measured squares
        px = 8192 * 4096
        W = 15
        unw = (st['4096']['alone']['unwrap_s_one_component_alone'] + st['8192']['alone']['unwrap_s_one_component_alone']) / 2 * 1.2
        name = '8192x4096'
    else:
        px = n * n
        W = st[str(n)]['windows']
        unw = st[str(n)]['alone']['unwrap_s_one_component_alone']
        name = '%d^2' % n
    tile = math.ceil(W / N) * tw
    if N == 1:
        gather = hand = 0.0
        t_stream = tile + 2 * unw
        t_step = tile + 2 * unw
    else:
        gather = 3 * 4 * px / N / LINK
        hand = 4 * px / LINK
        t_stream = tile + 2 * unw / N + gather + hand / (N / 2)
        t_step = tile + unw + 5 * 4 * px / N / LINK + 2 * 4 * px / LINK
    print('| %d | %s | %d | %.1f ms | %.1f ms | %.1f ms | %.1f ms | **%.1f ms** | **%.0f** | %.1f ms | %.0f |' % (
        N, name, W, tile * 1e3, 2 * unw / N * 1e3, gather * 1e3, hand / max(N / 2, 1) * 1e3 if N > 1 else 0, t_stream * 1e3,
        px / t_stream / 1e6, t_step * 1e3, px / t_step / 1e6))
