#!/usr/bin/env python
"""Print VGPR/AGPR/scratch/occupancy of selected kernels: python tools_resusage.py file.hip 'regex'"""
import re, subprocess, sys
src, pat = sys.argv[1], sys.argv[2]
r = subprocess.run(['hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-fno-gpu-rdc', '-ffp-contract=fast', '-fno-slp-vectorize',
                    '-Rpass-analysis=kernel-resource-usage', '-I', 'pygpa_amd/csrc', '-c', src, '-o', '/tmp/_res.o'], capture_output=True, text=True)
if r.returncode:
    print(r.stderr[-3000:]); sys.exit(1)
blocks = re.split(r'remark: [^\n]*Function Name: ', r.stderr)[1:]
for b in blocks:
    name = b.split()[0]
    dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    short = re.sub(r'\(.*', '', dn.replace('(anonymous namespace)::', '')).replace('void gpa::', '')
    if not re.search(pat, short):
        continue
    g = lambda k: (re.search(k + r': (\d+)', b) or [0, -1])[1]
    print('%-48s VGPR %3s AGPR %3s scratch %4s occ %s LDS %s' % (short, g('VGPRs'), g('AGPRs'), g(r'ScratchSize \[bytes/lane\]'),
          g(r'Occupancy \[waves/SIMD\]'), g(r'LDS Size \[bytes/block\]')))
