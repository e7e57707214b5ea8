#!/bin/bash
ulimit -c 0
cd /tmp && export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
for cfg in "512 64" "1024 16"; do set -- $cfg
  out=$ROOT/gpurun_out/stackprof_$1; rm -rf $out; mkdir -p $out
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $ROOT/tools/stack_bench.py --sizes $1 --stacks $2 --reps 10 > $out/log.txt 2>&1
  f=$(find $out -name '*kernel_stats.csv' | head -1); cp $f $ROOT/gpurun_out/stack_kernel_stats_$1.csv
  rm -rf $out
done
