#!/bin/bash
# usage (GPU box): TAG=name [REPS=2] [ARGS="--size 2048 --kgrid 4x2"] tools/ab_env.sh "VAR=1 VAR2=3" "VAR=2" ...   -> gpurun_out/ab_env_$TAG.txt
# bench.py (whole step + HIP-event kernel times of the sweep kernels) per set of library switches ("-" = none)
cd "$GRAFT_REPO_ROOT" || exit 1
run() { timeout 300 python3 bench.py --steps ${STEPS:-30} --warmup 4 --no-cpu --no-f64 --no-pipeline --no-config5 $ARGS 2>/dev/null | KERN="${KERN:-pass}" python3 -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%.1f Mpix/s  %.3f ms  resident %.1f ' % (d['value'], d['ms_per_step'], d['resident_only']['value']), {k: round(v['total_ms'],4) for k,v in d['kernels'].items() if any(s in k for s in os.environ['KERN'].split(','))})"; }
{
echo "ARGS=$ARGS"
for rep in $(seq 1 ${REPS:-2}); do
  for e in "$@"; do
    echo -n "rep $rep [$e]: "
    if [ "$e" = "-" ]; then run; else env $e bash -c "$(declare -f run); ARGS='$ARGS' STEPS='$STEPS' KERN='$KERN' run"; fi
  done
done
} | tee gpurun_out/ab_env_${TAG:-x}.txt
