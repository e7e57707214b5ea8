#!/bin/bash
out=gpurun_out/r2k; mkdir -p $out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py -q -m gpu -x 2>&1 | tail -4
for s in 512 1024 2048; do python bench.py --size $s --no-cpu --no-f64 --steps 50 > $out/bench_$s.json 2>> $out/bench.err; done
python - <<'PY'
import json
for s in (512,1024,2048):
    d=json.load(open('gpurun_out/r2k/bench_%d.json'%s)); print(s, d['value'], d['ms_per_step'], d['resident_only']['value'], d['stage_ms'])
PY
