#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for v in base mt128; do
  lib=""; [ "$v" != base ] && lib=$GRAFT_REPO_ROOT/pygpa_amd/variants/libgpa_$v.so
  echo "== $v"; GPA_HIP_LIB=$lib SIZES="1024 1500 2048" bash tools/gpu_sizes.sh
done
echo "== old"; ENVV=GPA_NO_SHARED=1 SIZES="1024 2048" bash tools/gpu_sizes.sh
