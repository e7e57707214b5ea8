#!/bin/bash
out=gpurun_out/triq; mkdir -p $out
for q in 0 1 2 4; do
  for s in 1000 1500 3000; do
    if [ $q = 0 ]; then unset GPA_TRI_Q; else export GPA_TRI_Q=$q; fi
    python bench.py --size $s --no-cpu --no-f64 --steps 10 > $out/b_${s}_$q.json 2>> $out/err
  done
done
python - <<'PY'
import json
for q in (0,1,2,4):
    for s in (1000,1500,3000):
        try:
            d=json.load(open('gpurun_out/triq/b_%d_%d.json'%(s,q))); print('Q',q,'size',s,d['value'],'colsolve',round(d['kernels']['colsolve_kernel']['avg_us_all_launches'],1))
        except Exception as e: print(q,s,'ERR',e)
PY
tail -2 $out/err
