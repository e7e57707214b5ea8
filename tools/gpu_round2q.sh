#!/bin/bash
out=gpurun_out/r2q; mkdir -p $out
timeout 2000 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $out/gpu_tests.log
cat $out/gpu_tests.log
for cfg in "4096 f64" "3000 f64" "1000 f64"; do
  set -- $cfg
  python bench.py --size $1 --dtype $2 --no-cpu --no-f64 --steps 5 > $out/bench_$1_$2.json 2>> $out/bench.err
done
python - <<'PY'
import json
for tag in ('4096_f64','3000_f64','1000_f64'):
    try:
        d=json.load(open('gpurun_out/r2q/bench_%s.json'%tag)); print(tag, d['value'], d['ms_per_step'], d['config']['unwrap_iters'])
        print('    ', ' '.join('%s %.1f' % (k.replace('_kernel',''), v['avg_us_all_launches']) for k,v in d['kernels'].items()))
    except Exception as e: print(tag,'ERR',e)
PY
tail -3 $out/bench.err
