#!/usr/bin/env python
"""Static instruction counts (total / VALU / LDS / VMEM / SALU / barriers) of the functions of one translation unit.
usage: python tools/isa_counts.py gpa_sweep [substring ...]   (substrings select demangled names)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
tu, subs = sys.argv[1], sys.argv[2:]
out = os.path.join(tempfile.gettempdir(), tu + '.s')
subprocess.run(['hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-fno-gpu-rdc', '-ffp-contract=fast', '-fno-slp-vectorize',
                '-Wno-unused-result', '-Wno-unused-value', '--cuda-device-only', '-S', os.path.join(ROOT, 'pygpa_amd', 'csrc', tu + '.hip'),
                '-o', out], capture_output=True)
txt = open(out).read()
rows = []
for m in re.finditer(r'\n(_Z\w+):[^\n]*\n(.*?)\n\.Lfunc_end', txt, re.S):
    ins = [l.strip() for l in m.group(2).split('\n') if re.match(r'\s+[a-z]', l) and not l.strip().startswith(('.', ';'))]
    rows.append((m.group(1), len(ins), sum(l.startswith('v_') for l in ins), sum(l.startswith('ds_') for l in ins),
                 sum(bool(re.match(r'(global_|buffer_|flat_|scratch_)', l)) for l in ins), sum(l.startswith('s_') for l in ins),
                 sum(l.startswith('s_barrier') for l in ins)))
names = subprocess.run(['c++filt'] + [r[0] for r in rows], capture_output=True, text=True).stdout.split('\n')
for r, d in zip(rows, names):
    d = d.replace('(anonymous namespace)::', '').split('(')[0].replace('void gpa::', '')
    if subs and not any(s in d for s in subs):
        continue
    print('%-56s total %5d valu %5d lds %4d vmem %4d salu %4d barriers %d' % ((d[:56],) + r[1:]))
