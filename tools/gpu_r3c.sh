#!/bin/bash
ulimit -c 0
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3c
timeout 900 python tools/check_shared_passb.py --sizes 1024x4096,512x2048,700x1024 --oracle > gpurun_out/r3c/check.txt 2>&1
cat gpurun_out/r3c/check.txt | tail -20
bash tools/gpu_variants.sh base nostore nofix pf 2>&1 | tee gpurun_out/r3c/variants.txt
BENCH_ARGS="--dtype f64" bash tools/gpu_variants.sh base 2>&1 | tee gpurun_out/r3c/variants_f64.txt
GPA_NO_SHARED=1 BENCH_ARGS="--dtype f64" bash tools/gpu_variants.sh base 2>&1 | tee -a gpurun_out/r3c/variants_f64.txt
