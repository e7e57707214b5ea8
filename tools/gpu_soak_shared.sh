#!/bin/bash
# soak of the shared-forward pass B: CASES random (row length, sigma, grid, peak) draws against the oracle
ulimit -c 0
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/soak; mkdir -p $out
export GPA_TEST_RANDOM_CASES=${CASES:-120} GPA_TEST_RANDOM_SEED=${SEED:-4242}
timeout 2400 python -m pytest tests/test_gpu_shared_passb.py -m gpu -x -q -k random_rows 2>&1 | tail -25 > $out/soak_shared_$GPA_TEST_RANDOM_SEED.log
cat $out/soak_shared_$GPA_TEST_RANDOM_SEED.log
