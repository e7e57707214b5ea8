#!/usr/bin/env python
"""VGPRs / scratch / occupancy / LDS of every kernel of one translation unit, from hipcc's kernel-resource-usage remarks.
usage: python tools/kernel_resources.py gpa_sweep [filter-substring] [extra hipcc flags...]"""
import re
import subprocess
import sys
import os

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
tu = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith('-') else ''
extra = [a for a in sys.argv[2:] if a.startswith('-')]
cmd = ['hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-fno-gpu-rdc', '-ffp-contract=fast', '-fno-slp-vectorize',
       '-Wno-unused-result', '-Wno-unused-value', '-Rpass-analysis=kernel-resource-usage', '-c',
       os.path.join(ROOT, 'pygpa_amd', 'csrc', tu + '.hip'), '-o', '/dev/null'] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r'Function Name: (\S+)', line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    for key, pat in (('vgpr', r' VGPRs: (\d+)'), ('agpr', r'AGPRs: (\d+)'), ('scratch', r'ScratchSize \[bytes/lane\]: (\d+)'),
                     ('occ', r'Occupancy \[waves/SIMD\]: (\d+)'), ('lds', r'LDS Size \[bytes/block\]: (\d+)'), ('sgpr', r' SGPRs: (\d+)')):
        m = re.search(pat, line)
        if m and cur:
            rows[cur][key] = int(m.group(1))
for name, r in rows.items():
    dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r'\(anonymous namespace\)::', '', dem).split('(')[0].replace('void gpa::', '')
    if flt and flt not in dem:
        continue
    print('%-70s vgpr %3d agpr %3d scratch %4d occ %d lds %6d' % (dem[:70], r.get('vgpr', -1), r.get('agpr', 0), r.get('scratch', -1),
                                                              r.get('occ', -1), r.get('lds', 0)))
