#!/bin/bash
# elements per thread of the unwrap transforms: 8 up to 2^10 (built-in), up to 2^11, up to 2^12 -- f32 and f64
out=gpurun_out/e8; mkdir -p $out
for v in default e8_11 e8_12; do
  lib=""; [ $v != default ] && lib=$PWD/tools/ubench/bin/libgpa_$v.so
  for s in 1024 2048 4096; do
    GPA_HIP_LIB=$lib python bench.py --size $s --no-cpu --steps 30 > $out/b_${v}_$s.json 2>> $out/err
  done
done
python - <<'PY'
import json
for v in ('default','e8_11','e8_12'):
    for s in (1024,2048,4096):
        try:
            d=json.load(open('gpurun_out/e8/b_%s_%d.json'%(v,s)))
            print(v,s,'f32',d['value'],d['resident_only']['value'],d['config']['unwrap_iters'],'f64',d.get('f64',{}).get('value'),d.get('f64',{}).get('unwrap_iters'))
        except Exception as e: print(v,s,'failed',e)
PY
tail -5 $out/err
