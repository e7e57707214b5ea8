#!/bin/bash
# usage: tools/pmc_run.sh <tag> <counter-set> [<counter-set> ...]   (each set = one rocprofv3 --pmc pass)
# Writes gpurun_out/pmc_<tag>_<i>/ ; collect with tools/pmc_summary.py
# Every pass runs under `timeout`: a counter set the hardware cannot collect makes rocprofv3 abort and
# then hang (seen: 25 minutes until gpurun's own limit).
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for set in "$@"; do
  timeout 240 rocprofv3 --pmc $set --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_$i.log 2>&1
  echo "pass $i ($set): rc=$?"
  i=$((i+1))
done
