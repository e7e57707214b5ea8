#!/bin/bash
# usage (GPU box): tools/ab_stag.sh -> gpurun_out/ab_stag.txt
# pass A phase stagger (PA_STAG=<phases>, PA_STAG_TICKS=<10-ns ticks per phase step>): the line owners of an x-plane start their
# plane loops a fraction of a plane period apart.  bench.py HIP-event kernel times + whole step, 4096^2 3 x 16 f32 (and f64).
cd "$GRAFT_REPO_ROOT" || exit 1
run() { timeout 300 python3 bench.py --steps 30 --warmup 4 --no-cpu --no-f64 --no-pipeline --no-config5 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%.1f Mpix/s  %.3f ms  resident %.1f ' % (d['value'], d['ms_per_step'], d['resident_only']['value']), {k: round(v['total_ms'],4) for k,v in d['kernels'].items() if 'pass' in k})"; }
{
for rep in 1 2; do
  echo -n "rep $rep base: "; run
  for ph in ${PHASES:-2 3 4}; do
    for tk in ${TICKS:-200 400 600 800 1200}; do
      echo -n "rep $rep PA_STAG=$ph TICKS=$tk: "; GPA_PA_STAG=$ph GPA_PA_STAG_TICKS=$tk run
    done
  done
done
echo "== f64"
echo -n "f64 base: "; run --dtype f64
for tk in 400 800 1600; do echo -n "f64 PA_STAG=2 TICKS=$tk: "; GPA_PA_STAG=2 GPA_PA_STAG_TICKS=$tk run --dtype f64; done
} | tee gpurun_out/ab_stag.txt
