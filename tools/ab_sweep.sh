#!/bin/bash
# usage (GPU box): tools/ab_sweep.sh -- pass A / pass B variants of round 5 on one box (bench.py, HIP-event kernel times)
cd "$GRAFT_REPO_ROOT" || exit 1
KERNEL=pass bash tools/gpu_variants.sh base pa_c2
echo "== GPA_PBS_E8=1 (shared pass B with eight elements per thread, 512 threads per row)"
for r in 1 2; do
GPA_PBS_E8=1 timeout 300 python3 bench.py --steps 10 --warmup 3 --no-cpu --no-f64 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('PBS_E8     %.1f Mpix/s  %.3f ms  resident %.1f ' % (d['value'], d['ms_per_step'], d['resident_only']['value']), {k: round(v['total_ms'],4) for k,v in d['kernels'].items() if 'pass' in k})"
done
