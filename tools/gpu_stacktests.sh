#!/bin/bash
ulimit -c 0
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py tests/test_gpu_configs.py -m gpu -q -k "stack" 2>&1 | tail -8
