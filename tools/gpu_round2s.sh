#!/bin/bash
out=gpurun_out/r2o; mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mixed_radix or transform_free or unwrap or fused_driver or random_shapes" 2>&1 | tail -15 > $out/tri.log
cat $out/tri.log
SIZES="${SIZES:-500 1000 1500 2000 3000}" bash tools/gpu_round2o.sh | tail -11
for cfg in "3000 f64" "1000 f64"; do
  set -- $cfg
  python bench.py --size $1 --dtype $2 --no-cpu --no-f64 --steps 5 > $out/bench_$1_$2.json 2>> $out/bench.err
done
python - <<'PY'
import json
for tag in ('3000_f64','1000_f64'):
    try:
        d=json.load(open('gpurun_out/r2o/bench_%s.json'%tag)); print(tag, d['value'], d['ms_per_step'], d['config']['unwrap_iters'])
        print('    ', ' '.join('%s %.1f' % (k.replace('_kernel',''), v['avg_us_all_launches']) for k,v in d['kernels'].items()))
    except Exception as e: print(tag,'ERR',e)
PY
