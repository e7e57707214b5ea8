#!/bin/bash
# several images in flight (one plan each), eager launches vs hipGraph replay
out=gpurun_out/inflight; mkdir -p $out
for g in 0 1; do
  for d in 1 4 8 16; do
    for s in 512 1024; do
      if [ $g = 1 ]; then export GPA_USE_GRAPH=1; else unset GPA_USE_GRAPH; fi
      timeout 300 python bench.py --size $s --no-cpu --no-f64 --steps 40 --inflight $d > $out/b_${g}_${d}_$s.json 2>> $out/err
    done
  done
done
python - <<'PY'
import json
for g in (0,1):
    for s in (512,1024):
        row=[]
        for d in (1,4,8,16):
            try:
                j=json.load(open('gpurun_out/inflight/b_%d_%d_%d.json'%(g,d,s))); row.append('D=%d: %.0f (%.0f resident)'%(d,j['value'],j['resident_only']['value']))
            except Exception as e: row.append('D=%d ERR'%d)
        print('graph' if g else 'eager', s, ' | '.join(row))
PY
tail -3 $out/err
