ulimit -c 0
ROOT=$GRAFT_REPO_ROOT; out=$ROOT/gpurun_out/salu; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $out/p -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-f64 --no-config5 > $out/p.log 2>&1
echo rc=$?
python3 - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$out/p/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        m = re.search(r'(\w+_kernel)', r['Kernel_Name'])
        k = m.group(1) if m else r['Kernel_Name'][:30]
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    g = lambda c: sum(d[c]) / max(len(d[c]), 1) if c in d else 0
    w = max(g('SQ_WAVES'), 1)
    print('%-28s n=%4d waves %9.0f per wave: VALU %7.0f SALU %7.0f SMEM %5.0f LDS %5.0f VMEMRD %5.0f WR %5.0f' % (k, len(d['SQ_WAVES']), w, g('SQ_INSTS_VALU') / w, g('SQ_INSTS_SALU') / w, g('SQ_INSTS_SMEM') / w, g('SQ_INSTS_LDS') / w, g('SQ_INSTS_VMEM_RD') / w, g('SQ_INSTS_VMEM_WR') / w))
PY
