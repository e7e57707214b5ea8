#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/tests
timeout 1800 python -m pytest tests/test_gpu_configs.py -q -m gpu -x -k "pipelined_stream or bench_gpus or two_ranks or nccl" > gpurun_out/tests/pytest_mgpu.log 2>&1
echo rc=$?; tail -15 gpurun_out/tests/pytest_mgpu.log
