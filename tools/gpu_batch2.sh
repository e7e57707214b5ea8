#!/bin/bash
out=gpurun_out/batch; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_dropin.py tests/test_abi.py -m gpu -x -q -k "stack or abi" 2>&1 | tail -5
python bench.py --no-cpu --steps 10 > $out/bench.json 2> $out/bench.err; tail -2 $out/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/batch/bench.json')); print(d['value'], d['ms_per_step'], d.get('small_image_stacks'))
PY
