#!/bin/bash
out=gpurun_out/tail; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py -m gpu -x -q 2>&1 | tail -3
for s in 256 512 1024 4096; do python bench.py --size $s --no-cpu --no-f64 --steps 40 > $out/b_$s.json 2>> $out/err; done
python - <<'PY'
import json
for s in (256,512,1024,4096):
    d=json.load(open('gpurun_out/tail/b_%d.json'%s)); print(s,d['value'],d['resident_only']['value'],d['config']['unwrap_iters'])
PY
timeout 300 python tools/stack_bench.py --sizes 512 --stacks 1,16,64 2>&1 | tail -4
