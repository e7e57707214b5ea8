#!/bin/bash
cd "$GRAFT_REPO_ROOT"
TESTS="tests/test_gpu_shared_passb.py" bash tools/gpu_tests.sh
for v in base padw2; do
  lib=""; [ "$v" != base ] && lib=$GRAFT_REPO_ROOT/pygpa_amd/variants/libgpa_$v.so
  echo "== $v"; GPA_HIP_LIB=$lib SIZES="1500 2000 3000" bash tools/gpu_sizes.sh
done
