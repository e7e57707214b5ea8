#!/bin/bash
out=gpurun_out/r2j; mkdir -p $out
for s in 500 512 1000 1024 1500 2000 2048 3000; do
  python bench.py --size $s --no-cpu --no-f64 --steps 20 > $out/bench_$s.json 2>> $out/bench.err
done
python - <<'PY'
import json
for s in (500,512,1000,1024,1500,2000,2048,3000):
    try:
        d=json.load(open('gpurun_out/r2j/bench_%d.json'%s)); print(s, d['value'], d['ms_per_step'], d['resident_only']['value'], d['config']['unwrap_iters'], d['stage_ms'])
    except Exception as e: print(s,'ERR',e)
PY
tail -5 $out/bench.err
