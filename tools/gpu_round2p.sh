#!/bin/bash
out=gpurun_out/r2o; mkdir -p $out
timeout 300 tools/ubench/bin/mrfft_bench > $out/mrfft_bench2.txt 2>&1; cat $out/mrfft_bench2.txt
bash tools/gpu_round2o.sh
