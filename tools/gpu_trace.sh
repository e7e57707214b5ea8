#!/bin/bash
# kernel timeline of one 512^2 step (rocprofv3 --kernel-trace): durations and gaps of the dependent chain
out=gpurun_out/trace512; mkdir -p $out
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $ROOT/$out/kt -- python3 $ROOT/bench.py --size ${1:-512} --steps 6 --warmup 2 --no-cpu --no-f64 > $ROOT/$out/log 2>&1
cd $ROOT
f=$(ls $out/kt/*/*kernel_trace.csv | head -1)
python3 - $f <<'PY'
import csv,sys,re
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last complete step: find the last passA and take from the preceding mean kernel up to the next mean
idx=[i for i,r in enumerate(rows) if 'passA_kernel' in r['Kernel_Name']]
i0=idx[-3]; i1=idx[-2]
seg=rows[i0-2:i1-2]
t0=int(seg[0]['Start_Timestamp'])
prev_end=None
tot=0
for r in seg:
    n=re.sub(r'\(.*','',r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void gpa::',''))[:34]
    s=int(r['Start_Timestamp'])-t0; e=int(r['End_Timestamp'])-t0
    gap = (s-prev_end) if prev_end is not None else 0
    if len(sys.argv) > 2: print('%-36s start %8.2f us  dur %6.2f  gap %6.2f  grid %s wg %s' % (n, s/1e3, (e-s)/1e3, gap/1e3, r.get('Grid_Size_X','?'), r.get('Workgroup_Size_X','?')))
    prev_end=e
print('segment wall: %.1f us, kernels %d' % ((int(seg[-1]['End_Timestamp'])-t0)/1e3, len(seg)))
import collections
d=collections.defaultdict(list)
for r in seg:
    n=re.sub(r'\(.*','',r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void gpa::',''))[:34]
    d[n].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for n,v in d.items():
    v.sort(); print('SUMMARY %-36s n=%3d median %.2f us  min %.2f  max %.2f' % (n,len(v),v[len(v)//2],v[0],v[-1]))
PY
rm -rf $out/kt
