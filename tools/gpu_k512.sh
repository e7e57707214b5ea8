#!/bin/bash
out=gpurun_out/k512; mkdir -p $out
for s in 512 1024; do
python bench.py --size $s --no-cpu --no-f64 --steps 40 > $out/b_$s.json 2>> $out/err
python - <<PY
import json
d=json.load(open('gpurun_out/k512/b_$s.json'))
print($s, d['value'], d['resident_only']['value'])
print(d['stage_ms'])
for k,v in d['kernels'].items(): print('  %-28s launches %3d  total %.4f ms  avg %.2f us' % (k, v['launches'], v['total_ms'], v['avg_us_all_launches']))
PY
done
