#!/bin/bash
ulimit -c 0
timeout 180 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "latency_tuned or driver_vs_oracle_512 or fused_driver_golden or stack_equals or a7_unwrap_golden or iteration_counts or rectangular" 2>&1 | tail -4
for s in 128 256 512 1024; do for off in 1 ""; do GPA_NO_ROWPQ=$off timeout 100 python bench.py --size $s --no-cpu --no-f64 --steps 40 | python -c "import json,sys; d=json.load(sys.stdin); print('norowpq=$off', d['config']['workload'][:12], d['value'], d['resident_only']['value'], d['config']['unwrap_iters'])"; done; done
