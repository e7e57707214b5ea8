#!/bin/bash
# round 3, first look at the shared-forward pass B: correctness vs the per-candidate kernel / oracle, then the bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3a
timeout 900 python tools/check_shared_passb.py --sizes 1024x4096,512x2048,4096x4096 --oracle > gpurun_out/r3a/check.txt 2>&1
tail -20 gpurun_out/r3a/check.txt
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu > gpurun_out/r3a/bench_shared.json 2> gpurun_out/r3a/bench_shared.err
GPA_NO_SHARED=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu > gpurun_out/r3a/bench_old.json 2> gpurun_out/r3a/bench_old.err
python - <<'PY'
import json
for n in ('shared', 'old'):
    try:
        d = json.loads(open('gpurun_out/r3a/bench_%s.json' % n).read().strip().split('\n')[-1])
        print(n, d['value'], d['ms_per_step'], {k: v.get('total_ms') for k, v in d.get('kernels', {}).items()}, d.get('f64', {}).get('value'))
    except Exception as e:
        print(n, 'failed', e)
PY
