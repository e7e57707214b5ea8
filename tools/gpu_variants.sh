#!/bin/bash
# usage (GPU box): [BENCH_ARGS="--size 2048 --kgrid 4x2"] [KERNEL=passB] tools/gpu_variants.sh base name1 name2 ...
# bench.py per library variant built by tools/variant.sh ("base" = the shipped library): Mpix/s, ms per step and the
# HIP-event time of the kernels whose name contains $KERNEL
ulimit -c 0
cd "$GRAFT_REPO_ROOT" || exit 1
args="${BENCH_ARGS:---no-f64}"
kern="${KERNEL:-passB}"
for v in "$@"; do
  lib=""; [ "$v" != base ] && lib=$GRAFT_REPO_ROOT/pygpa_amd/variants/libgpa_$v.so
  for r in 1 2; do
    GPA_HIP_LIB=$lib timeout 300 python3 bench.py --steps 10 --warmup 3 --no-cpu $args 2>/dev/null | KERN=$kern V=$v python3 -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-10s %.1f Mpix/s  %.3f ms  resident %.1f ' % (os.environ['V'], d['value'], d['ms_per_step'], d['resident_only']['value']), {k: round(v['total_ms'],4) for k,v in d['kernels'].items() if os.environ['KERN'] in k}, d['config']['unwrap_iters'])"
  done
done
