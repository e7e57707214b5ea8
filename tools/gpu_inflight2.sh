#!/bin/bash
out=gpurun_out/inflight; mkdir -p $out
for q in 4 8 16 24; do
  export GPU_MAX_HW_QUEUES=$q
  for d in 1 4 8; do
    for s in 512 1024; do
      timeout 300 python bench.py --size $s --no-cpu --no-f64 --steps 40 --inflight $d > $out/q_${q}_${d}_$s.json 2>> $out/err
    done
  done
done
python - <<'PY'
import json
for q in (4,8,16,24):
    for s in (512,1024):
        row=[]
        for d in (1,4,8):
            try:
                j=json.load(open('gpurun_out/inflight/q_%d_%d_%d.json'%(q,d,s))); row.append('D=%d: %.0f (%.0f resident)'%(d,j['value'],j['resident_only']['value']))
            except Exception as e: row.append('D=%d ERR'%d)
        print('hwq',q, s, ' | '.join(row))
PY
tail -3 $out/err
