#!/bin/bash
out=gpurun_out/r2f; mkdir -p $out
python tools/enqueue_cost.py 2>&1 | tee $out/enqueue_cost.txt
GPA_NO_WORKER=1 python tools/enqueue_cost.py 512 1024 2>&1 | sed 's/^/noworker /' | tee -a $out/enqueue_cost.txt
python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py -q -m gpu -x 2>&1 | tail -4
python -m pytest tests/test_gpu_configs.py -q -m gpu -x -k "not two_ranks and not 16384 and not 8192" 2>&1 | tail -4
for s in 512 1024 2048 4096; do python bench.py --size $s --no-cpu --no-f64 --steps 50 > $out/bench_$s.json 2>> $out/bench.err; done
python - <<'PY'
import json
for s in (512,1024,2048,4096):
    d=json.load(open('gpurun_out/r2f/bench_%d.json'%s)); print(s, d['value'], d['ms_per_step'], d['resident_only']['value'])
PY
