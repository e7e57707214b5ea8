#!/bin/bash
out=gpurun_out/batch; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "image_stack" 2>&1 | tail -15 > $out/test.log
cat $out/test.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py -m gpu -x -q 2>&1 | tail -4
timeout 600 python tools/stack_bench.py --sizes 256,512,1024,500,1000 --stacks 1,4,8,16,32,64 2>&1 | tee $out/stack_bench3.txt
