#!/bin/bash
# usage (GPU box): COMMIT=<sha> [SKIP_TESTS=1] [SKIP_PMC=1] tools/gpu_close.sh
# Closing pass of a round: the whole GPU suite, smoke, rocprofv3 kernel stats (4096^2 f32 / f64, 3000^2, 512^2)
# the PMC passes behind profiles/counters.json (one pass per counter set: the guide's rule), then the bench line (which
# reads those counters).  Everything lands under
# gpurun_out/close/; copy what is to be kept into profiles/ with the round's prefix.
ulimit -c 0
ROOT=$GRAFT_REPO_ROOT
cd "$ROOT" || exit 1
out=$ROOT/gpurun_out/close; mkdir -p $out
if [ -z "$SKIP_TESTS" ]; then
  timeout 2400 python -m pytest tests -q -m gpu -x --durations=12 > $out/pytest_gpu.log 2>&1
  echo "pytest rc=$?"; tail -4 $out/pytest_gpu.log
fi
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
# (the two unwrap components run one after the other while the kernels are timed: concurrent kernels of the two streams
#  stretch each other's durations)
export GPA_SERIAL_UNWRAP=1
kstats() {   # tag, bench args...
  tag=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ks_$tag -- python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu --no-f64 "$@" > $out/ks_$tag.log 2>&1
  f=$(ls $out/ks_$tag/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp $f $out/kernel_stats_$tag.csv && head -12 $out/kernel_stats_$tag.csv | cut -c1-150
  rm -rf $out/ks_$tag
}
kstats 4096_f32
kstats 4096_f64 --dtype f64
kstats 3000_f32 --size 3000
kstats 2048_c2 --size 2048 --kgrid 4x2
kstats 512_f32 --size 512
unset GPA_SERIAL_UNWRAP
# one unwrap component at 8192^2 / 16384^2 and the tile pipeline's image stream at 16384^2 (-> gpurun_out/kstats/)
WHAT="unwrap8192 unwrap16384 tiles16384 lf16384" bash $ROOT/tools/gpu_kstats.sh > $out/kstats_long.log 2>&1; tail -3 $out/kstats_long.log | cut -c1-150
cp $ROOT/gpurun_out/kstats/kernel_stats_unwrap_8192.csv $ROOT/gpurun_out/kstats/kernel_stats_unwrap_16384.csv $ROOT/gpurun_out/kstats/kernel_stats_tiles_16384.csv $ROOT/gpurun_out/kstats/kernel_stats_lf_16384.csv $out/ 2>/dev/null
cd $ROOT && timeout 600 python3 tools/lf_times.py --sizes 2048 4096 8192 16384 > $out/lf_times.txt 2>&1; GPA_NO_LFTILE=1 timeout 600 python3 tools/lf_times.py --sizes 4096 16384 > $out/lf_times_rowkernel.txt 2>&1; cd /tmp
cp $ROOT/gpurun_out/stage_times.json $out/ 2>/dev/null
# round 6: the rows either side of the path -- the resident image -> k-vectors -> u -> undistortion -> properties leg with its
# rocprofv3 summary, the same in f64, the NumPy-in / NumPy-out calls of a9 / f-2 / f-3 / f-4 under rocprofv3, the step's timeline
cd $ROOT && bash tools/gpu_pipeline.sh 4096 f32 > $out/pipeline_4096_f32.txt 2>&1; bash tools/gpu_pipeline.sh 4096 f64 > $out/pipeline_4096_f64.txt 2>&1
cp gpurun_out/pipeline/kernel_stats_pipeline_4096_f32.csv gpurun_out/pipeline/kernel_stats_pipeline_4096_f64.csv gpurun_out/pipeline/pipeline_4096_f32.json gpurun_out/pipeline/pipeline_4096_f64.json $out/ 2>/dev/null
WHAT="next4096 next2048 next16384" NEXT_ARGS="--what per peaks deconv plane" bash tools/gpu_kstats.sh > $out/kstats_next.log 2>&1
cp gpurun_out/kstats/kernel_stats_next_4096.csv gpurun_out/kstats/kernel_stats_next_2048.csv gpurun_out/kstats/kernel_stats_next_16384.csv $out/ 2>/dev/null
grep -h "\^2 f" gpurun_out/kstats/ks_next_*.log > $out/next_rows_host_calls.txt 2>/dev/null
bash tools/gpu_timeline.sh 4096 f32 > /dev/null 2>&1; bash tools/gpu_timeline.sh 4096 f64 > /dev/null 2>&1; cp gpurun_out/timeline/timeline_4096_f32.txt gpurun_out/timeline/timeline_4096_f64.txt $out/ 2>/dev/null
cd /tmp
cd $ROOT && timeout 600 bash tools/gpu_unwrap_sizes.sh > /dev/null 2>&1; cp gpurun_out/unwrap_sizes.txt $out/ 2>/dev/null
SIZES="256 500 512 1000 1024 1500 2000 2048 3000 4096 8192 16384" timeout 900 bash tools/gpu_sizes.sh > /dev/null 2>&1; cp gpurun_out/sizes.txt $out/ 2>/dev/null
cd /tmp
bench_line() {   # the bench line LAST: its roofline reads the counters of this very tree (profiles/counters.json, checked by hash)
  cd $ROOT; timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; tail -2 $out/bench.err; head -c 700 $out/bench.json; echo
}
if [ -n "$SKIP_PMC" ]; then bench_line; exit 0; fi
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_WAVES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"; do
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $out/pmc_$i -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-f64 > $out/pmc_$i.log 2>&1
  echo "pmc pass $i ($set): rc=$?"
  i=$((i+1))
done
cd $ROOT
COMMIT=${COMMIT:-unknown} python3 tools/make_counters.py $out/counters.json $out/pmc_[0-9] > /dev/null; head -c 900 $out/counters.json; echo
cp $out/counters.json $ROOT/profiles/counters.json
# HBM traffic of the Lawler-Fujita kernels at 16384^2 (one pass per counter, as above)
j=0
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $out/pmclf_$j -- python3 $ROOT/tools/lf_times.py --sizes 16384 --reps 1 > $out/pmclf_$j.log 2>&1
  j=$((j+1))
done
python3 tools/pmc_summary.py $out/pmclf_[0-9] > $out/lf_pmc_16384.txt 2>&1
rm -rf $out/pmclf_[0-9]
# f64 traffic of the sweep (18.2 GB moved for 4 GB needed)
j=0
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $out/pmc64_$j -- python3 $ROOT/bench.py --dtype f64 --steps 2 --warmup 1 --no-cpu --no-f64 > $out/pmc64_$j.log 2>&1
  j=$((j+1))
done
COMMIT=${COMMIT:-unknown} PMC_DTYPE=f64 python3 tools/make_counters.py $out/counters_f64.json $out/pmc64_[0-9] > /dev/null
cp $out/counters_f64.json $ROOT/profiles/counters_f64.json
rm -rf $out/pmc_[0-9] $out/pmc64_[0-9]
bench_line
