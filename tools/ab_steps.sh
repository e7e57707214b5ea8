cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for cfg in "20 3" "50 5" "100 10" "20 20"; do
  set -- $cfg
  echo -n "rep $rep steps $1 warmup $2: "; python bench.py --steps $1 --warmup $2 --no-cpu --no-f64 --no-pipeline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.1f Mpix/s %7.3f ms  resident %8.1f' % (d['value'], d['ms_per_step'], d['resident_only']['value']))"
done; done
