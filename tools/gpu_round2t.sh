#!/bin/bash
# round 2, closing pass: whole GPU suite, bench line, kernel stats, PMC passes -> counters; kernel stats of a 3000^2 run
ulimit -c 0
out=gpurun_out/r2t; mkdir -p $out
timeout 1200 python -m pytest tests -q -m gpu -x --durations=12 > $out/pytest_gpu.log 2>&1
tail -22 $out/pytest_gpu.log
python __graft_entry__.py smoke 2>&1 | tail -1
python bench.py > $out/bench.json 2> $out/bench.err; tail -3 $out/bench.err; cat $out/bench.json
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$out/kstats -- python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu --no-f64 > $ROOT/$out/kstats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$out/kstats3000 -- python3 $ROOT/bench.py --size 3000 --steps 5 --warmup 1 --no-cpu --no-f64 > $ROOT/$out/kstats3000.log 2>&1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_WAVES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"; do
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $ROOT/$out/pmc_$i -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-f64 > $ROOT/$out/pmc_$i.log 2>&1
  echo "pmc pass $i ($set): rc=$?"
  i=$((i+1))
done
cd $ROOT
python3 tools/make_counters.py $out/counters.json $out/pmc_* > /dev/null; head -c 1500 $out/counters.json
f=$(ls $out/kstats/*/*kernel_stats.csv | head -1); cp $f $out/kernel_stats.csv; head -16 $out/kernel_stats.csv | cut -c1-160
f=$(ls $out/kstats3000/*/*kernel_stats.csv | head -1); cp $f $out/kernel_stats_3000.csv; head -16 $out/kernel_stats_3000.csv | cut -c1-160
rm -rf $out/kstats $out/kstats3000 $out/pmc_[0-9]
