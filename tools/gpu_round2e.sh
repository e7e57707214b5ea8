#!/bin/bash
out=gpurun_out/r2e; mkdir -p $out
python -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -6
python bench.py --no-cpu --dtype f64 --steps 5 > $out/bench_f64_tri.json 2>> $out/bench.err
GPA_COLSOLVE=fft python bench.py --no-cpu --dtype f64 --steps 5 > $out/bench_f64_fft.json 2>> $out/bench.err
python bench.py --no-cpu --no-f64 > $out/bench.json 2>> $out/bench.err
GPA_COLSOLVE=tri python bench.py --no-cpu --no-f64 > $out/bench_f32_tri.json 2>> $out/bench.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2e/bench*.json')):
    try:
        d=json.load(open(f)); print(f, d['value'], d['ms_per_step'], d['resident_only']['value'], d['config']['unwrap_iters'], d['stage_ms'], {k:round(v['total_ms'],3) for k,v in d['kernels'].items() if 'col' in k or 'row' in k})
    except Exception as e: print(f,'ERR',e)
PY
