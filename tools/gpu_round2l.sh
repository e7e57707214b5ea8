#!/bin/bash
# compact padded mode: parity + sizes
mkdir -p gpurun_out/r2l
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r2l/parity.log
cat gpurun_out/r2l/parity.log
bash tools/gpu_sizes.sh > gpurun_out/r2l/sizes.txt 2>&1
cat gpurun_out/r2l/sizes.txt | tail -20
