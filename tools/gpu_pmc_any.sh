#!/bin/bash
# usage: TAG=name ARGS="--size 4096 --dtype f64" bash tools/gpu_pmc_any.sh : FETCH / WRITE per kernel of a bench run
out=gpurun_out/pmc_$TAG; mkdir -p $out
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $ROOT/$out/pmc_$i -- python3 $ROOT/bench.py $ARGS --steps 2 --warmup 1 --no-cpu --no-f64 > $ROOT/$out/pmc_$i.log 2>&1
  echo "pmc pass $i ($set): rc=$?"
  i=$((i+1))
done
cd $ROOT
python3 - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$out/pmc_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r'(\w+_kernel)', r['Kernel_Name'])
        if m: acc[m.group(1)][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in sorted(acc.items()):
    row = {}
    for c, v in cs.items():
        top = max(v); w = [x for x in v if x > 0.05 * top] or v
        row[c] = sum(w) / len(w)
    if 'FETCH_SIZE' in row and 'WRITE_SIZE' in row:
        print('%-28s read %8.1f MB  write %8.1f MB  total %8.1f MB' % (k, 2 * row['FETCH_SIZE'] / 1024, row['WRITE_SIZE'] / 1024, (2 * row['FETCH_SIZE'] + row['WRITE_SIZE']) / 1024))
PY
rm -rf $out/pmc_[0-9]
