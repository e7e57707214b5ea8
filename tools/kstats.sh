#!/bin/bash
# usage (on the GPU box): tools/kstats.sh <tag> [bench args] -> gpurun_out/ks_<tag>.txt with per-kernel averages, serial unwrap
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
GPA_SERIAL_UNWRAP=1 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ks_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu "$@" > /dev/null 2>&1
python3 - <<PY > $GRAFT_REPO_ROOT/gpurun_out/ks_$tag.txt
import csv,glob
f=glob.glob('$GRAFT_REPO_ROOT/gpurun_out/ks_$tag/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print('%-46s calls %4s avg_us %9.1f' % (r['Name'].replace('void gpa::','').replace('(anonymous namespace)::','')[:46], r['Calls'], float(r['AverageNs'])/1e3))
PY
cat $GRAFT_REPO_ROOT/gpurun_out/ks_$tag.txt
