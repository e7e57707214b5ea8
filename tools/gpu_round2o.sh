#!/bin/bash
# mixed-radix unwrap: parity of the affected tests + kernel breakdown at non-power-of-two sizes
out=gpurun_out/r2o; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mixed_radix or unwrap or fused_driver or random_shapes" 2>&1 | tail -5 > $out/mr.log
cat $out/mr.log
for s in ${SIZES:-500 1000 1500 3000}; do
  python bench.py --size $s --no-cpu --no-f64 --steps 10 > $out/bench_$s.json 2>> $out/bench.err
done
python - <<'PY'
import json, os
for s in os.environ.get('SIZES', '500 1000 1500 3000').split():
    try:
        d=json.load(open('gpurun_out/r2o/bench_%s.json'%s)); print(s, d['value'], d['ms_per_step'], d['config']['unwrap_iters'])
        print('    ', ' '.join('%s %.1f' % (k.replace('_kernel',''), v['avg_us_all_launches']) for k,v in d['kernels'].items()))
    except Exception as e: print(s,'ERR',e)
PY
tail -3 $out/bench.err
