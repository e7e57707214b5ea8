#!/bin/bash
ulimit -c 0
timeout 180 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "transform_free or mixed_radix or chirpz or random or stack or golden or elongated" 2>&1 | tail -3
for s in 500 600 1000; do for small in 0 640; do GPA_TRI_SMALL=$small timeout 100 python bench.py --size $s --no-cpu --no-f64 --steps 30 | python -c "import json,sys; d=json.load(sys.stdin); print('small<=$small', d['config']['workload'][:12], d['value'], d['resident_only']['value'], d['config']['unwrap_iters'])"; done; done
for small in 0 640; do GPA_TRI_SMALL=$small timeout 100 python bench.py --size 512 --dtype f64 --no-cpu --no-f64 --steps 30 | python -c "import json,sys; d=json.load(sys.stdin); print('f64 small<=$small', d['config']['workload'][:12], d['value'], d['resident_only']['value'], d['config']['unwrap_iters'])"; done
