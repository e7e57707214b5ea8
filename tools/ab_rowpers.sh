cd $GRAFT_REPO_ROOT
for v in "" "GPA_NO_ROWPERS=1" "GPA_PAIR_MAXSIDE=4096" "GPA_PAIR_MAXSIDE=4096 GPA_NO_ROWPERS=1" "GPA_SERIAL_UNWRAP=1" "GPA_SERIAL_UNWRAP=1 GPA_NO_ROWPERS=1"; do
  echo "== $v"
  env $v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu --no-f64 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], 'resident', d['resident_only']['value'], 'iters', d['config']['unwrap_iters'], 'unwrap serial', d['stage_ms'].get('unwrap(serial, both components)'))"
done
