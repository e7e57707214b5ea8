#!/bin/bash
# soak: widened randomized parity sweeps (not part of the default suite)
ulimit -c 0
out=gpurun_out/soak; mkdir -p $out
export GPA_TEST_RANDOM_CASES=${CASES:-250} GPA_TEST_RANDOM_SEED=${SEED:-4242}
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "random_smooth or random_shapes_unwrap or random_shapes_driver or random_stacks" 2>&1 | tail -15 > $out/soak_$GPA_TEST_RANDOM_SEED.log
cat $out/soak_$GPA_TEST_RANDOM_SEED.log
