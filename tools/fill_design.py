#!/usr/bin/env python3
"""Fill the [[placeholders]] of DESIGN.md (and print README's table values) from the committed closing pass under profiles/:
r06_bench.json, r06_kernel_stats_*.csv, r06_timeline_4096_f32.txt, r06_unwrap_sizes.txt, r06_sizes.txt, r06_lf_times.txt.
usage: python tools/fill_design.py [--check]   (--check: only report placeholders that would stay)"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = lambda *a: os.path.join(ROOT, 'profiles', *a)
d = json.load(open(P('r06_bench.json')))
K = d['kernels']


def stats(name):
    out = {}
    for r in csv.DictReader(open(P(name))):
        nm = re.sub(r'^void ', '', r['Name'])
        nm = re.sub(r'gpa::(\(anonymous namespace\)::)?', '', nm)
        nm = nm.split('<')[0].split('(')[0]
        e = out.setdefault(nm, [0, 0.0])
        e[0] += int(r['Calls'])
        e[1] += float(r['TotalDurationNs'])
    return {k: (c, t / c / 1e3) for k, (c, t) in out.items()}   # calls, average us


ks = stats('r06_kernel_stats_4096_f32.csv')
what = {
    'passA_kernel': 'image once + 12 complex x-planes out: 1.68 GB',
    'passB_shared_kernel': '12 x-planes in + 3 raw winners and indices out: 2.01 GB',
    'reconstruct_setup_kernel': '3 lock-ins in, wnorm + 2 r0 out: 0.60 GB',
    'rowdct_fused_kernel': 'first iteration only: r in, R out',
    'pqdct_kernel': 'p, w in, D = DCT_rows(q) out: 201 MB',
    'colstream_agg_kernel': 'R, D in, R out (+ chunk sums): 188 MB',
    'colstream_scan_kernel': 'chunk carries: 8.8 MB',
    'colstream_apply_kernel': 'R in, Z out: 134 MB',
    'rowidct_p_kernel': 'Z, p in, p out: 201 MB',
    'pq_kernel': 'first iteration only: p, w in, q out',
    'phi_flush_kernel': '10 kept p in, phi out: 0.72 GB',
}
bound = {
    'passA_kernel': 'VALU at 2 waves per SIMD (0.47 ms) + the drain of 32-byte row pieces (0.21 ms)',
    'passB_shared_kernel': 'VALU: wave-instructions x 4 cycles = the duration (3 waves per SIMD, LDS-bound occupancy)',
    'reconstruct_setup_kernel': 'VALU (atan2 / sqrt / two 2x2 solves per pixel: 470 instructions per wave and row)',
    'pqdct_kernel': 'latency: four rows of p and w per row pair, two memory round trips, 4 waves per SIMD',
    'colstream_agg_kernel': 'HBM', 'colstream_apply_kernel': 'HBM', 'rowidct_p_kernel': 'HBM (persistent, LDS-DMA)',
    'colstream_scan_kernel': 'latency (0.9 MB of carries)', 'phi_flush_kernel': 'HBM', 'pq_kernel': 'HBM', 'rowdct_fused_kernel': 'latency',
}
rocname = {'rowidct_p_kernel': 'rowidct_p_pers_kernel'}
rows = ['| kernel | must move (algorithmic) | counter bytes | us (rocprofv3, working launch) | GB/s on algorithmic bytes (of 8 TB/s) | VALU wave-instructions x 4 cycles / duration | bound by |', '|---|---|---|---|---|---|---|']
cnt = json.load(open(P('counters.json')))
for k in ['passA_kernel', 'passB_shared_kernel', 'reconstruct_setup_kernel', 'pqdct_kernel', 'colstream_agg_kernel', 'colstream_scan_kernel',
          'colstream_apply_kernel', 'rowidct_p_kernel', 'phi_flush_kernel', 'pq_kernel', 'rowdct_fused_kernel']:
    v = K[k]
    us = ks.get(rocname.get(k, k), (0, v['avg_us_all_launches']))[1]
    alg = v.get('algorithmic_bytes_per_launch') or v.get('hbm_bytes_per_launch')
    gbps = alg / us / 1e3
    c = cnt.get(k, {})
    hb = c.get('hbm_bytes')
    vi = c.get('valu_insts')
    v4 = (vi * 4 / (1024 * 2.4e9) / (us * 1e-6)) if vi else None
    rows.append('| `%s` | %s | %s | %.1f | %.0f (%.2f) | %s | %s |' % (k, what.get(k, ''), ('%.0f MB' % (hb / 1e6)) if hb and hb < 1e9 else (('%.2f GB' % (hb / 1e9)) if hb else 'n/a'),
                                                                     us, gbps, gbps / 8000, ('%.2f' % v4) if v4 else 'n/a', bound.get(k, '')))
kernel_table = '\n'.join(rows)

# pipeline table
pe = d['pipeline_end_to_end']
prow = ['| stage | kernel | launches | ms (HIP events) | compulsory GB | of 8 TB/s |', '|---|---|---|---|---|---|']
for st, kk in pe['kernels'].items():
    for kn, v in kk.items():
        if kn.startswith('('):
            continue
        prow.append('| %s | `%s` | %d | %.4f | %s | %s |' % (st, kn, v['launches'], v['total_ms'], v.get('compulsory_GB', ''), v.get('frac_of_hbm_peak', '')))
next_table = '\n'.join(prow)
sm = pe['stage_ms']
pipe_stages = ', '.join('%s %.2f' % (k.replace('wfr2_grad_opt_x3+phasegradient2J+props_from_Jac', '`wfr2_grad_opt` x 3 + `phasegradient2J` + `props_from_Jac`'), v) for k, v in sm.items()) + ' ms'
pk = pe['kernels']['extract_primary_ks']
fft_rows = pk['dft_rows_r2c_kernel']['total_ms'] * 1e3
fft_cols = pk['dft_cols_kernel']['total_ms'] * 1e3 + pk.get('dft_rows_kernel', {'total_ms': 0})['total_ms'] * 1e3
peaks_ms = sum(v['total_ms'] for v in pk.values())

tl = open(P('r06_timeline_4096_f32.txt')).read()
m = re.findall(r'step \d+: wall .*? ([\d.]+) us \| GPU busy ([\d.]+) us .*?>= 2 kernels in flight ([\d.]+) us.*?\n.*?\n\s+sweep phase ([\d.]+) us \| unwrap phase ([\d.]+) us', tl)
wall, busy, two, sweep, unw = [float(x) for x in m[-1]]
unwrap_bytes = 16.0e9
f64 = d['f64']
ru = d['roofline_unwrap']
us_sz = open(P('r06_unwrap_sizes.txt')).read()


def solve_ms(n):
    mm = re.search(r'%d\^2 f32 default\s+([\d.]+) ms/solve.*?rowdct_fused ([\d.]+) us ([\d.]+);.*?rowidct_p ([\d.]+) us ([\d.]+);' % n, us_sz)
    return mm.groups() if mm else None


s14, s13 = solve_ms(16384), solve_ms(8192)
vals = {
    'value': '%.0f' % d['value'], 'ms': '%.2f' % d['ms_per_step'], 'f64_mpix': '%.0f' % f64['value'], 'f64_ms': '%.2f' % f64['ms_per_step'],
    'c2_mpix': '%.0f' % d['config2']['value'], 'c5_mpix': '%.0f' % d['config5_single_gpu']['value'],
    'pipe_ms': '%.1f' % pe['ms_per_image'], 'pipe_mpix': '%.0f' % pe['value'], 'pipe_stages': pipe_stages,
    'fft_us': '%.0f' % (fft_rows + fft_cols), 'rows_us': '%.0f' % fft_rows, 'cols_us': '%.0f' % fft_cols, 'cols_us_full': '%.0f' % (2 * fft_cols),
    'peaks_ms': '%.2f' % peaks_ms, 'peaks_nodog_ms': '%.2f' % (peaks_ms - pk.get('gauss_fft_cols_kernel', {'total_ms': 0})['total_ms'] - pk.get('gauss_fft_rows_kernel', {'total_ms': 0})['total_ms']),
    'two_frac': '%.0f' % (100 * two / unw), 'unwrap_ms': '%.2f' % (unw / 1e3), 'unwrap_tbs': '%.1f' % (unwrap_bytes / (unw * 1e-6) / 1e12),
    'unwrap_frac': '%.2f' % (unwrap_bytes / (unw * 1e-6) / 6.3e12), 'sweep_ms': '%.2f' % (sweep / 1e3),
    'kernel_table': kernel_table, 'next_table': next_table,
    'iter_us': '%.0f' % sum(ks.get(rocname.get(k, k), (0, 0))[1] for k in ru['group']),
    'passA_ms': '%.2f' % (ks['passA_kernel'][1] / 1e3), 'passB_ms': '%.2f' % (ks['passB_shared_kernel'][1] / 1e3), 'setup_ms': '%.2f' % (ks['reconstruct_setup_kernel'][1] / 1e3),
    'unw_serial_ms': '%.2f' % ru['unwrap_serial_ms_both_components'], 'roof_frac': '%.2f' % d['roofline']['frac'],
    'design_kb': '%d' % (os.path.getsize(os.path.join(ROOT, 'DESIGN.md')) // 1024),
}
if s14:
    vals.update({'hp_solve14': s14[0], 'hp_rowdct14': '%.2f' % (float(s14[1]) / 1e3), 'hp_rowidct14': '%.2f' % (float(s14[3]) / 1e3), 'hp_rowidct14_frac': s14[4]})
if s13:
    vals['hp_solve13'] = s13[0]
for k, v in list(vals.items()):
    if len(sys.argv) > 1 and sys.argv[1] == '--show' and '\n' not in v:
        print(k, '=', v)
path = os.path.join(ROOT, 'DESIGN.md')
s = open(path).read()
for k, v in vals.items():
    s = s.replace('[[' + k + ']]', v)
left = sorted(set(re.findall(r'\[\[(\w+)\]\]', s)))
print('placeholders left:', left)
if '--check' not in sys.argv and '--show' not in sys.argv:
    open(path, 'w').write(s)
