#!/bin/bash
# usage (GPU box): [SIZES="4096 8192 16384"] [MODES="default tri fft stream"] [DTYPE=f32] [KMAX=10] tools/gpu_unwrap_sizes.sh
# One weighted unwrap component (gpa_unwrap_prediff_dev on resident data) per size and column-solve mode: ms per solve,
# iterations, and the HIP-event time of every kernel per working launch -> gpurun_out/unwrap_sizes.txt
ulimit -c 0
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout ${T:-900} python3 tools/unwrap_sizes.py --sizes ${SIZES:-4096 8192 16384} --modes ${MODES:-default tri fft} --dtype ${DTYPE:-f32} --kmax ${KMAX:-10} $EXTRA 2>&1 | tee gpurun_out/unwrap_sizes${TAG:+_$TAG}.txt
