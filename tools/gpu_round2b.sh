#!/bin/bash
# round 2, pass b: whole GPU suite, bench, kernel stats, PMC passes -> counters
out=gpurun_out/r2i; mkdir -p $out
python -m pytest tests -q -m gpu -x --durations=12 > $out/pytest_gpu.log 2>&1
tail -25 $out/pytest_gpu.log
python bench.py > $out/bench.json 2> $out/bench.err; tail -3 $out/bench.err; cat $out/bench.json
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$out/kstats -- python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu --no-f64 > $ROOT/$out/kstats.log 2>&1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_WAVES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"; do
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $ROOT/$out/pmc_$i -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-f64 > $ROOT/$out/pmc_$i.log 2>&1
  echo "pmc pass $i ($set): rc=$?"
  i=$((i+1))
done
cd $ROOT
python3 tools/make_counters.py $out/counters.json $out/pmc_* > /dev/null; cat $out/counters.json | head -80
f=$(ls $out/kstats/*/*kernel_stats.csv | head -1); cp $f $out/kernel_stats.csv; head -25 $out/kernel_stats.csv | cut -c1-200
# keep the merged output small
rm -rf $out/kstats $out/pmc_[0-9]
