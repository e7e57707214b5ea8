#!/bin/bash
# usage (GPU box): [WHAT="bench4096 unwrap8192 unwrap16384 tiles16384 lf16384"] tools/gpu_kstats.sh
# rocprofv3 --kernel-trace --stats of: the bench step at 4096^2 (unwrap components one after the other), one unwrap component
# at 8192^2 / 16384^2 (tools/unwrap_sizes.py), the tile pipeline's image stream at 16384^2 (tools/stage_times.py), the Lawler-Fujita
# undistortion at 16384^2 (tools/lf_times.py)
# -> gpurun_out/kstats/kernel_stats_<what>.csv
ulimit -c 0
ROOT=$GRAFT_REPO_ROOT
out=$ROOT/gpurun_out/kstats; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() {   # tag, program args...
  tag=$1; shift
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ks_$tag -- python3 "$@" > $out/ks_$tag.log 2>&1
  f=$(ls $out/ks_$tag/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp $f $out/kernel_stats_$tag.csv && head -${LINES:-14} $out/kernel_stats_$tag.csv | cut -c1-160
  rm -rf $out/ks_$tag
}
for w in ${WHAT:-bench4096 unwrap8192 unwrap16384 tiles16384}; do
  case $w in
    bench4096) GPA_SERIAL_UNWRAP=1 run 4096_f32 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu --no-f64 ;;
    bench4096f64) GPA_SERIAL_UNWRAP=1 run 4096_f64 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu --no-f64 --dtype f64 ;;
    unwrap*) n=${w#unwrap}; run unwrap_$n $ROOT/tools/unwrap_sizes.py --sizes $n --modes default --reps 2 ;;
    tiles*) n=${w#tiles}; run tiles_$n $ROOT/tools/stage_times.py --sizes $n ;;
    lf*) n=${w#lf}; run lf_$n $ROOT/tools/lf_times.py --sizes $n --reps 2 ;;
    next*) n=${w#next}; run next_$n $ROOT/tools/next_rows.py --sizes $n ${NEXT_ARGS:-} ;;
  esac
done
