#!/bin/bash
# usage (GPU box): [N=16384] bash tools/lf_pmc.sh -- instruction / wait counters of the Lawler-Fujita kernels (one counter set per pass)
ulimit -c 0
ROOT=$GRAFT_REPO_ROOT; out=$ROOT/gpurun_out/lfpmc; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "TCP_TA_TCP_STATE_READ_sum TA_BUSY_avr TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $out/p$i -- python3 $ROOT/tools/lf_times.py --sizes ${N:-16384} --reps 1 > $out/p$i.log 2>&1
  echo "pass $i rc=$?"
  i=$((i+1))
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$out/p*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void gpa::(anonymous namespace)::', '')
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print('   %-28s n=%d mean %.4g' % (c, len(v), sum(v) / len(v)))
PY
