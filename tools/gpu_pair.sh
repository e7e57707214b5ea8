#!/bin/bash
out=gpurun_out/pair; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py -m gpu -x -q 2>&1 | tail -4
for s in 256 512 1024 500 1000; do
  python bench.py --size $s --no-cpu --no-f64 --steps 40 > $out/p_$s.json 2>> $out/err
  GPA_NO_PAIR=1 python bench.py --size $s --no-cpu --no-f64 --steps 40 > $out/n_$s.json 2>> $out/err
done
python - <<'PY'
import json
for s in (256,512,1024,500,1000):
    a=json.load(open('gpurun_out/pair/p_%d.json'%s)); b=json.load(open('gpurun_out/pair/n_%d.json'%s))
    print(s, 'paired', a['value'], a['resident_only']['value'], a['config']['unwrap_iters'], '| two streams', b['value'], b['resident_only']['value'], b['config']['unwrap_iters'])
PY
tail -2 $out/err
