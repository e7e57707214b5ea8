#!/bin/bash
# transform-free column solve at small power-of-two sizes (f32): is its shorter instruction stream faster there?
out=gpurun_out/trismall; mkdir -p $out
for s in 256 512 1024 2048; do
  for mode in d t; do
    GPA_COLSOLVE=$mode python bench.py --size $s --no-cpu --no-f64 --steps 40 > $out/b_${s}_$mode.json 2>> $out/err
  done
done
python - <<'PY'
import json
for s in (256,512,1024,2048):
    r=[]
    for m in 'dt':
        d=json.load(open('gpurun_out/trismall/b_%d_%s.json'%(s,m))); r.append((d['value'],d['resident_only']['value'],d['config']['unwrap_iters']))
    print(s,'dct',r[0],'tri',r[1])
PY
