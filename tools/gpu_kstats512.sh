#!/bin/bash
# rocprofv3 kernel stats of the 512^2 single-image bench (final build of round 2)
ulimit -c 0
out=gpurun_out/k512s; mkdir -p $out
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$out/ks -- python3 $ROOT/bench.py --size 512 --steps 40 --warmup 3 --no-cpu --no-f64 > $ROOT/$out/log 2>&1
cd $ROOT
f=$(ls $out/ks/*/*kernel_stats.csv | head -1); cp $f $out/kernel_stats_512.csv; head -14 $out/kernel_stats_512.csv | cut -c1-150
rm -rf $out/ks
