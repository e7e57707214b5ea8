#!/bin/bash
out=gpurun_out/batch; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "image_stack" 2>&1 | tail -15 > $out/test.log
cat $out/test.log
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py -m gpu -x -q 2>&1 | tail -4
