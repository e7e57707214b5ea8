#!/bin/bash
out=gpurun_out/r2x; mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "chirpz or mixed_radix or random_smooth or random_shapes_unwrap or golden" 2>&1 | tail -15 > $out/chirpz.log
cat $out/chirpz.log
for s in 1004 1392 3004; do
  python bench.py --size $s --no-cpu --no-f64 --steps 10 > $out/bench_$s.json 2>> $out/bench.err
  GPA_NO_MR=1 python bench.py --size $s --no-cpu --no-f64 --steps 10 > $out/bench_${s}_nomr.json 2>> $out/bench.err
done
python - <<'PY'
import json
for s in ('1004','1004_nomr','1392','1392_nomr','3004','3004_nomr'):
    try:
        d=json.load(open('gpurun_out/r2x/bench_%s.json'%s)); print(s, d['value'], d['ms_per_step'], d['config']['unwrap_iters'])
        print('    ', ' '.join('%s %.1f' % (k.replace('_kernel',''), v['avg_us_all_launches']) for k,v in d['kernels'].items()))
    except Exception as e: print(s,'ERR',e)
PY
tail -3 $out/bench.err
