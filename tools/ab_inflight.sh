#!/bin/bash
# usage (GPU box): tools/ab_inflight.sh  -> gpurun_out/ab_inflight.txt
# images in flight (bench.py --inflight: one plan = sweep stream + two unwrap streams per image in flight) against the occupancy of
# the sweep's row pass (PBS_LDS_PAD: extra LDS per workgroup -> 3 / 2 / 1 rows per CU): can the VALU-bound sweep of image i + 1
# share the CUs with the HBM-bound unwrap of image i?
cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 60 --warmup 4 --no-cpu --no-f64 --no-pipeline "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mpix/s %7.3f ms  resident %8.1f %7.3f ms' % (d['value'], d['ms_per_step'], d['resident_only']['value'], d['resident_only']['ms_per_step']))"; }
for rep in 1 2; do
for pad in 0 9000 20000 34000; do
  for inf in 1 2 3 4; do
    echo -n "rep $rep PBS_LDS_PAD=$pad inflight=$inf: "; GPA_PBS_LDS_PAD=$pad run --inflight $inf
  done
done
done | tee gpurun_out/ab_inflight.txt
