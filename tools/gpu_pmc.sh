#!/bin/bash
# usage (GPU box): TAG=name [ARGS="--size 4096 --dtype f64"] [LIB=variant] [ENVV="GPA_NO_SHARED=1"] tools/gpu_pmc.sh "SET 1 counters" "SET 2 counters" ...
# One rocprofv3 --pmc pass of bench.py per counter set (separate passes: the guide's rule), then per-kernel means of
# every counter -> gpurun_out/pmc_$TAG.txt.  Every pass runs under `timeout` (a counter set the hardware cannot
# collect makes rocprofv3 hang).
ulimit -c 0
ROOT=$GRAFT_REPO_ROOT
out=$ROOT/gpurun_out/pmc_$TAG; mkdir -p $out
[ -n "$LIB" ] && export GPA_HIP_LIB=$ROOT/pygpa_amd/variants/libgpa_$LIB.so
[ -n "$ENVV" ] && export $ENVV
cd /tmp && export TMPDIR=/tmp
i=0
for set in "$@"; do
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $out/pass_$i -- python3 $ROOT/bench.py $ARGS --steps 2 --warmup 1 --no-cpu --no-f64 > $out/pass_$i.log 2>&1
  echo "pmc pass $i ($set): rc=$?"
  i=$((i+1))
done
python3 $ROOT/tools/pmc_summary.py $out/pass_* > $ROOT/gpurun_out/pmc_$TAG.txt
rm -rf $out
grep -A12 -i "${KERNEL:-passB}" $ROOT/gpurun_out/pmc_$TAG.txt | head -60
