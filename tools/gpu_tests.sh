#!/bin/bash
# usage (GPU box): [TESTS="tests/test_gpu_shared_passb.py"] [PYTEST_ARGS="-k foo"] [NOX=1] tools/gpu_tests.sh  -- the GPU suite (or a part), log under gpurun_out/
ulimit -c 0
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/tests
xflag=-x; [ -n "$NOX" ] && xflag=""   # NOX=1: run on after a failure
timeout ${TEST_TIMEOUT:-2400} python -m pytest ${TESTS:-tests} -q -m gpu $xflag --durations=8 $PYTEST_ARGS > gpurun_out/tests/pytest_gpu.log 2>&1
echo "pytest rc=$?"
tail -${TAIL:-25} gpurun_out/tests/pytest_gpu.log
