#!/bin/bash
# usage (GPU box): [SIZES="500 1000 1500 2000 3000"] [ENVV="GPA_NO_SHARED=1"] tools/gpu_sizes.sh -- bench.py per image size (one image per call,
# D2H included / resident), with the per-kernel times of the sweep
ulimit -c 0
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; out=gpurun_out/sizes${TAG:+_$TAG}.txt; : > $out
[ -n "$ENVV" ] && export $ENVV
for n in ${SIZES:-500 512 1000 1024 1500 2000 2048 3000 4096}; do
  timeout 300 python3 bench.py --size $n --steps 10 --warmup 3 --no-cpu --no-f64 2>/dev/null | N=$n python3 -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels']
print('%5s^2  %8.1f Mpix/s  resident %8.1f  ' % (os.environ['N'], d['value'], d['resident_only']['value']), {n[:14]: round(v['total_ms']*1e3) for n,v in k.items() if 'pass' in n or 'recon' in n}, d['config']['unwrap_iters'])" | tee -a $out
done
