#!/bin/bash
out=gpurun_out/stackprof; mkdir -p $out
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$out/k -- python3 $ROOT/tools/stack_bench.py --sizes 512 --stacks 64 --reps 5 > $ROOT/$out/log.txt 2>&1
cd $ROOT
f=$(ls $out/k/*/*kernel_stats.csv | head -1); cp $f $out/kernel_stats_stack512x64.csv; head -14 $out/kernel_stats_stack512x64.csv | cut -c1-150
tail -3 $out/log.txt
rm -rf $out/k
