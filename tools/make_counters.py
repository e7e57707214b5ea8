#!/usr/bin/env python
"""profiles/counters.json from rocprofv3 --pmc csv output dirs (tools/pmc_run.sh):
    python tools/make_counters.py OUT.json gpurun_out/pmc_<tag>_*
Per kernel (short name as bench.py's profiler uses): HBM bytes per WORKING launch = 2 x FETCH_SIZE + WRITE_SIZE
(KB x 1024; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950), VALU wave-instructions, L2
requests.  Launches that returned at once (PCG iterations after convergence) are left out: a launch counts as
working when its counter is above 5 % of the kernel's largest."""
import collections
import csv
import glob
import json
import os
import sys

SHORT = ['passA_kernel', 'passB_shared_kernel', 'passB_kernel', 'reconstruct_setup_kernel', 'rowdct_fused_kernel', 'rowdct_half_kernel',
         'rowidct_p_half_kernel', 'pqdct_kernel', 'colstream_agg_kernel', 'colstream_scan_kernel', 'colstream_apply_kernel', 'colsolve_tri_kernel', 'colsolve_kernel',
         'rowidct_p_kernel', 'pq_kernel', 'phi_flush_kernel', 'mean_partial_kernel']
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[2:]:
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            kname = r['Kernel_Name'].replace('rowidct_p_pers_kernel', 'rowidct_p_kernel')   # (the persistent form files under the profiler's name)
            for s in SHORT:
                if s in kname:
                    if s in ('passB_kernel', 'passB_shared_kernel') and 'float' not in r['Kernel_Name']:
                        continue
                    acc[s][r['Counter_Name']].append(float(r['Counter_Value']))
                    break
out = {}
for k, cs in acc.items():
    row = {}
    m = {}
    for c, v in cs.items():
        top = max(v)
        w = [x for x in v if x > 0.05 * top] or v
        m[c] = sum(w) / len(w)
        row['n_' + c] = len(w)
    if 'FETCH_SIZE' in m and 'WRITE_SIZE' in m:
        row['hbm_bytes'] = int((2 * m['FETCH_SIZE'] + m['WRITE_SIZE']) * 1024)
        row['fetch_bytes_x2'] = int(2 * m['FETCH_SIZE'] * 1024)
        row['write_bytes'] = int(m['WRITE_SIZE'] * 1024)
    if 'SQ_INSTS_VALU' in m:
        row['valu_insts'] = int(m['SQ_INSTS_VALU'])
    if 'TCC_HIT_sum' in m and 'TCC_MISS_sum' in m:
        row['l2_requests'] = int(m['TCC_HIT_sum'] + m['TCC_MISS_sum'])
        row['l2_hit_rate'] = round(m['TCC_HIT_sum'] / max(m['TCC_HIT_sum'] + m['TCC_MISS_sum'], 1), 4)
    for c in ('SQ_WAIT_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_INSTS_VALU_MFMA_F32', 'SQ_INSTS_VMEM_WR', 'SQ_INSTS_VMEM_RD', 'SQ_WAVES', 'SQ_INSTS_LDS', 'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE', 'SQ_WAVE_CYCLES', 'SQ_WAIT_INST_ANY', 'SQ_BUSY_CYCLES', 'SQ_ACTIVE_INST_VALU', 'GRBM_GUI_ACTIVE'):
        if c in m:
            row[c] = int(m[c])
    out[k] = row
out['_note'] = ('per WORKING launch, rocprofv3 --pmc separate passes of `python bench.py --no-cpu --no-f64` (4096^2, 3x16, f32); '
                'hbm_bytes = (2 x FETCH_SIZE + WRITE_SIZE) KB x 1024 (FETCH_SIZE doubled per MI355X_MICROARCH.md for gfx950)')
import datetime
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from bench import csrc_sha256   # the kernel sources these counters belong to: bench.py refuses them on any other tree
out['_meta'] = {'commit': os.environ.get('COMMIT', 'unknown'), 'csrc_sha256': csrc_sha256(), 'date': datetime.date.today().isoformat(),
                'config': {'n': int(os.environ.get('PMC_SIZE', '4096')), 'K': int(os.environ.get('PMC_K', '16')),
                           'dtype': os.environ.get('PMC_DTYPE', 'f32')},
                'how': 'rocprofv3 --pmc, one pass per counter set, bench.py --steps 2 --warmup 1 --no-cpu --no-f64'}
json.dump(out, open(sys.argv[1], 'w'), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
