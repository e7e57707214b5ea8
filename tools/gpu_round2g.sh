#!/bin/bash
out=gpurun_out/r2g; mkdir -p $out
ROOT=$PWD
for s in 512 1024; do
  for d in 1 2 4 8; do python bench.py --size $s --no-cpu --no-f64 --steps 100 --inflight $d > $out/bench_${s}_d$d.json 2>> $out/bench.err; done
done
python bench.py --size 4096 --no-cpu --no-f64 --inflight 2 > $out/bench_4096_d2.json 2>> $out/bench.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2g/bench_*.json')):
    try:
        d=json.load(open(f)); print(f, d['value'], d['ms_per_step'], d['resident_only']['value'])
    except Exception as e: print(f,'ERR',e)
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$out/ks512 -- python3 $ROOT/bench.py --size 512 --steps 20 --warmup 2 --no-cpu --no-f64 > $ROOT/$out/ks512.log 2>&1
cd $ROOT
f=$(ls $out/ks512/*/*kernel_stats.csv | head -1); cp $f $out/kernel_stats_512.csv; cut -d, -f1-4 $out/kernel_stats_512.csv | cut -c1-60,200- | head -14
t=$(ls $out/ks512/*/*kernel_trace.csv | head -1)
python3 - "$t" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# take a window of one step in the middle: print kernel name, start offset, duration, queue
mid=len(rows)//2
base=int(rows[mid]['Start_Timestamp'])
for r in rows[mid:mid+60]:
    print('%8.2f us  dur %6.2f us  q%s  %s' % ((int(r['Start_Timestamp'])-base)/1e3, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, r.get('Queue_Id','?'), r['Kernel_Name'].replace('void gpa::','').replace('(anonymous namespace)::','')[:40]))
PY
rm -rf $out/ks512
