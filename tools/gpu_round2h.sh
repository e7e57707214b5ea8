#!/bin/bash
out=gpurun_out/r2h; mkdir -p $out
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "transform_free or fused or golden or unwrap or random" 2>&1 | tail -4
GPA_COLSOLVE=tri python bench.py --no-cpu --no-f64 > $out/bench_f32_tri.json 2>> $out/bench.err
python bench.py --no-cpu --no-f64 > $out/bench_f32_fft.json 2>> $out/bench.err
python bench.py --no-cpu --dtype f64 --steps 5 > $out/bench_f64_tri.json 2>> $out/bench.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2h/bench*.json')):
    d=json.load(open(f)); print(f, d['value'], d['ms_per_step'], d['resident_only']['value'], {k:round(v['total_ms'],3) for k,v in d['kernels'].items() if 'col' in k})
PY
