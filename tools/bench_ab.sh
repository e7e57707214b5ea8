#!/bin/bash
# usage (GPU box): A="GPA_X=1" B="" [ARGS="--size 2048 --kgrid 4x2"] [REPS=2] tools/bench_ab.sh
# the bench line (no CPU / f64 legs) with environment A and with environment B, alternating, on ONE box: value, resident, early stop
# and the HIP-event time of the sweep kernels -- the boxes differ by ~3 %, an A/B has to share one
ulimit -c 0
cd "$GRAFT_REPO_ROOT" || exit 1
one() {
  env $1 timeout 300 python3 bench.py --no-cpu --no-f64 $ARGS 2>/dev/null | TAG="$1" python3 -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('%-28s value %7.1f  resident %7.1f  early %7.1f |' % (os.environ['TAG'] or '(default)', d['value'], d['resident_only']['value'], d.get('early_stop', {}).get('value', 0)),
      ' '.join('%s %.0f' % (n.replace('_kernel',''), v['total_ms']*1e3) for n,v in k.items() if 'pass' in n))"
}
for r in $(seq ${REPS:-2}); do one "$A"; one "$B"; done
