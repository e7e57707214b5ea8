#!/bin/bash
mkdir -p gpurun_out/r2n
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mixed_radix" 2>&1 | tail -25 > gpurun_out/r2n/mr.log
cat gpurun_out/r2n/mr.log
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py -m gpu -q 2>&1 | tail -8 > gpurun_out/r2n/parity.log
cat gpurun_out/r2n/parity.log
for s in 500 1000 1500 2000 3000; do
  python bench.py --size $s --no-cpu --no-f64 --steps 10 > gpurun_out/r2n/bench_$s.json 2>> gpurun_out/r2n/bench.err
done
python - <<'PY'
import json
for s in (500,1000,1500,2000,3000):
    try:
        d=json.load(open('gpurun_out/r2n/bench_%d.json'%s)); print(s, d['value'], d['ms_per_step'], d['config']['unwrap_iters'])
        for k,v in d['kernels'].items(): print('   ',k,v['launches'],round(v['total_ms'],3),round(v['avg_us_all_launches'],1))
    except Exception as e: print(s,'ERR',e)
PY
tail -3 gpurun_out/r2n/bench.err
