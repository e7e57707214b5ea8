#!/bin/bash
# usage (GPU box): tools/ab_tilemap.sh -> gpurun_out/ab_tilemap.txt
# pass A experiments: tile -> XCD map (PA_STAG=-q: groups of q lines dealt round-robin over the XCDs) x plane rotation (PA_ROT=m)
cd "$GRAFT_REPO_ROOT" || exit 1
run() { timeout 300 python3 bench.py --steps 30 --warmup 4 --no-cpu --no-f64 --no-pipeline --no-config5 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%.1f Mpix/s  %.3f ms  resident %.1f ' % (d['value'], d['ms_per_step'], d['resident_only']['value']), {k: round(v['total_ms'],4) for k,v in d['kernels'].items() if 'pass' in k})"; }
{
for rep in 1 2; do
  for q in ${QS:-0 1 2}; do for m in ${ROTS:-0 1 3 5}; do
    echo -n "rep $rep PA_STAG=-$q PA_ROT=$m $EXTRA: "
    if [ $q = 0 ]; then GPA_PA_ROT=$m run $EXTRA; else GPA_PA_STAG=-$q GPA_PA_ROT=$m run $EXTRA; fi
  done; done
done
} | tee gpurun_out/ab_tilemap${TAG}.txt
