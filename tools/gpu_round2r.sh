#!/bin/bash
out=gpurun_out/r2o; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q -k "compact or sweep or lockin or random_shapes_driver or golden or nccl" 2>&1 | tail -5 > $out/padded.log
cat $out/padded.log
SIZES="${SIZES:-500 1000 2000 3000}" bash tools/gpu_round2o.sh | tail -9
