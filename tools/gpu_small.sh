#!/bin/bash
# usage (GPU box): [SIZES="256 500 512 1000 1024"] tools/gpu_small.sh -- one image per call: Mpix/s and the per-kernel HIP-event times (us, launches)
ulimit -c 0
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; out=gpurun_out/small${TAG:+_$TAG}.txt; : > $out
for n in ${SIZES:-256 500 512 1000 1024}; do
  timeout 300 python3 bench.py --size $n --steps 40 --warmup 5 --no-cpu --no-f64 2>/dev/null | N=$n python3 -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels']
print('%5s^2  %8.1f Mpix/s  %.3f ms  resident %8.1f (%.3f ms)' % (os.environ['N'], d['value'], d['ms_per_step'], d['resident_only']['value'], d['resident_only']['ms_per_step']), d['config']['unwrap_iters'])
print('        ', {n.replace('_kernel','')[:16]: (v['launches'], round(v['total_ms']*1e3)) for n,v in k.items()})" | tee -a $out
done
