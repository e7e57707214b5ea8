#!/bin/bash
out=gpurun_out/r2d; mkdir -p $out
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "transform_free or fused or golden or unwrap" 2>&1 | tail -15
python bench.py --no-cpu --no-f64 > $out/bench.json 2> $out/bench.err; tail -3 $out/bench.err
GPA_COLSOLVE_FFT=1 python bench.py --no-cpu --no-f64 > $out/bench_fft.json 2>> $out/bench.err
for s in 512 1024 2048; do python bench.py --size $s --no-cpu --no-f64 --steps 50 > $out/bench_$s.json 2>> $out/bench.err; done
python bench.py --no-cpu --dtype f64 --steps 5 > $out/bench_f64.json 2>> $out/bench.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2d/bench*.json')):
    try:
        d=json.load(open(f)); print(f, d['value'], d['ms_per_step'], d['resident_only']['value'], d['config']['unwrap_iters'], {k:round(v['total_ms'],3) for k,v in d['kernels'].items() if 'col' in k or 'row' in k})
    except Exception as e: print(f,'ERR',e)
PY
