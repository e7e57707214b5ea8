#!/bin/bash
out=gpurun_out/r2c; mkdir -p $out
./tools/ubench/bin/coltile > $out/coltile.txt 2>&1; cat $out/coltile.txt
for s in 512 1024; do python bench.py --size $s --no-cpu --no-f64 --steps 50 > $out/bench_$s.json 2>> $out/bench.err; done
python - <<'PY'
import json
for s in (512,1024):
    d=json.load(open('gpurun_out/r2c/bench_%d.json'%s)); print(s, d['value'], d['ms_per_step'], d['resident_only']['value'], {k:v['total_ms'] for k,v in d['kernels'].items()})
PY
python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py -q -m gpu -x 2>&1 | tail -5
