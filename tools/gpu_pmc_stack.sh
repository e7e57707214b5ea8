#!/bin/bash
# usage (GPU box): [SIZE=512] [STACK=64] tools/gpu_pmc_stack.sh "SET 1" "SET 2" ... -- PMC passes of tools/stack_bench.py (one per counter set), per-kernel means -> gpurun_out/pmc_stack_$SIZE.txt
ulimit -c 0
ROOT=$GRAFT_REPO_ROOT
size=${SIZE:-512}; stack=${STACK:-64}
out=$ROOT/gpurun_out/pmc_stack_$size; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "$@"; do
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $out/pass_$i -- python3 $ROOT/tools/stack_bench.py --sizes $size --stacks $stack --reps 3 > $out/pass_$i.log 2>&1
  echo "pmc pass $i ($set): rc=$?"
  i=$((i+1))
done
python3 $ROOT/tools/pmc_summary.py $out/pass_* > $ROOT/gpurun_out/pmc_stack_$size.txt
rm -rf $out
grep -A14 "${KERNEL:-passB_kernel}" $ROOT/gpurun_out/pmc_stack_$size.txt | head -80
