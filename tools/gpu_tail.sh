#!/bin/bash
ulimit -c 0   # a faulting kernel must not spend the GPU budget on a core dump
out=gpurun_out/tail; mkdir -p $out
timeout 180 python __graft_entry__.py smoke 2>&1 | tail -2 || { echo SMOKE FAILED; exit 1; }
timeout 120 python bench.py --size 512 --no-cpu --no-f64 --steps 5 > $out/first.json 2>> $out/err || { echo FIRST BENCH FAILED; tail -5 $out/err; exit 1; }
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py -m gpu -x -q 2>&1 | tail -3
for s in 256 512 1024 2048 4096; do timeout 120 python bench.py --size $s --no-cpu --no-f64 --steps 40 > $out/b_$s.json 2>> $out/err; done
python - <<'PY'
import json
for s in (256,512,1024,2048,4096):
    d=json.load(open('gpurun_out/tail/b_%d.json'%s)); print(s,d['value'],d['resident_only']['value'],d['config']['unwrap_iters'])
PY
timeout 300 python tools/stack_bench.py --sizes 512 --stacks 1,16,64 2>&1 | tail -4
