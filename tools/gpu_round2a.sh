#!/bin/bash
# first GPU pass of round 2: new config tests, bench (graphs on/off), small sizes
out=gpurun_out/r2a; mkdir -p $out
python -m pytest tests/test_gpu_configs.py -q -m gpu --durations=15 > $out/configs.log 2>&1
tail -40 $out/configs.log
python bench.py --no-cpu > $out/bench.json 2> $out/bench.err; tail -3 $out/bench.err; cat $out/bench.json
for s in 512 1024 2048; do
  python bench.py --size $s --no-cpu --no-f64 --steps 50 > $out/bench_$s.json 2>> $out/bench.err
  GPA_NO_GRAPH=1 python bench.py --size $s --no-cpu --no-f64 --steps 50 > $out/bench_${s}_nograph.json 2>> $out/bench.err
done
GPA_NO_GRAPH=1 python bench.py --no-cpu --no-f64 > $out/bench_nograph.json 2>> $out/bench.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2a/bench*.json')):
    try:
        d=json.load(open(f)); print(f, d['value'], d['ms_per_step'], d['resident_only']['value'], d.get('f64',{}).get('value'))
    except Exception as e: print(f, 'ERR', e)
PY
