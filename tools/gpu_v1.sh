#!/bin/bash
out=gpurun_out/v1; mkdir -p $out
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py -m gpu -x -q 2>&1 | tail -12 > $out/test.log; cat $out/test.log
for s in 250 1001 1250; do
  python bench.py --size $s --no-cpu --no-f64 --steps 10 > $out/b_$s.json 2>> $out/err
  GPA_NO_MR=1 python bench.py --size $s --no-cpu --no-f64 --steps 10 > $out/n_$s.json 2>> $out/err
done
python - <<'PY'
import json
for s in (250,1001,1250):
    for t in ('b','n'):
        try:
            d=json.load(open('gpurun_out/v1/%s_%d.json'%(t,s))); print(s, 'fused' if t=='b' else 'old  ', d['value'], d['ms_per_step'], d['config']['unwrap_iters'])
        except Exception as e: print(s,t,'ERR',e)
PY
tail -2 $out/err
