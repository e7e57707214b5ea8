#!/bin/bash
# usage (GPU box): [BENCH_ARGS=...] tools/ab.sh name1 name2 ...   ("base" = the shipped library)
# prints Mpix/s + stage_ms of bench.py for each variant built by tools/variant.sh, and serial-unwrap kernel averages
args="$BENCH_ARGS"   # e.g. BENCH_ARGS="--size 2048 --kgrid 4x2" tools/ab.sh base
for v in "$@"; do
  lib=""; [ "$v" != base ] && lib=$GRAFT_REPO_ROOT/pygpa_amd/variants/libgpa_$v.so
  echo "== $v"
  for r in 1 2; do
    GPA_HIP_LIB=$lib python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu $args | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  %.1f Mpix/s  %.3f ms ' % (d['value'], d['ms_per_step']), {k[:10]: round(v,3) for k,v in d['stage_ms'].items()})"
  done
  export GPA_HIP_LIB=$lib
  bash $GRAFT_REPO_ROOT/tools/kstats.sh ab_$v $args | head -9 | sed 's/^/  /'
  unset GPA_HIP_LIB
done
