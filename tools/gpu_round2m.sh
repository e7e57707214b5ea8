#!/bin/bash
mkdir -p gpurun_out/r2m
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r2m/parity.log
cat gpurun_out/r2m/parity.log
for s in 1000 3000; do
  python bench.py --size $s --no-cpu --no-f64 --steps 10 > gpurun_out/r2m/bench_$s.json 2>> gpurun_out/r2m/bench.err
done
python - <<'PY'
import json
for s in (1000,3000):
    d=json.load(open('gpurun_out/r2m/bench_%d.json'%s)); print(s, d['value'], d['ms_per_step'], d['config']['unwrap_iters'])
    for k,v in d['kernels'].items(): print('   ',k,v['launches'],round(v['total_ms'],3),round(v['avg_us_all_launches'],1))
PY
tail -3 gpurun_out/r2m/bench.err
