#!/bin/bash
# usage (GPU box): tools/gpu_timeline.sh [size]  -> gpurun_out/timeline/timeline_<n>.txt
# kernel timeline of the benchmark step (rocprofv3 --kernel-trace, the two unwrap streams as they really run): GPU-busy time
# (union of the kernel intervals), time with >= 2 kernels in flight, the idle gaps, per-phase walls
n=${1:-4096}; dt=${2:-f32}
out=gpurun_out/timeline; mkdir -p $out
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $ROOT/$out/kt -- python3 $ROOT/bench.py --size $n --dtype $dt --steps 8 --warmup 2 --no-cpu --no-f64 --no-pipeline > $ROOT/$out/log 2>&1
cd $ROOT
f=$(ls $out/kt/*/*kernel_trace.csv | head -1)
python3 - $f > $out/timeline_${n}_$dt.txt <<'PY'
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
name = lambda r: re.sub(r'[<(].*', '', r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void gpa::', ''))
idx = [i for i, r in enumerate(rows) if name(r) == 'passA_kernel']
# steps of the timed loop with the D2H: take three consecutive steps from the middle of the run
mid = len(idx) // 3
for k in range(mid, mid + 3):
    a, b = idx[k], idx[k + 1]
    seg = rows[a:b]
    t0 = int(seg[0]['Start_Timestamp'])
    ev = []
    for r in seg:
        ev.append((int(r['Start_Timestamp']) - t0, 1)); ev.append((int(r['End_Timestamp']) - t0, -1))
    ev.sort()
    busy = two = 0; depth = 0; last = 0; gaps = []
    for t, d in ev:
        if depth >= 1: busy += t - last
        if depth >= 2: two += t - last
        if depth == 0 and t > last and last > 0: gaps.append((t - last, last))
        depth += d; last = t
    wall = int(rows[b]['Start_Timestamp']) - t0
    ser = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg)
    print('step %d: wall (passA to next passA) %.1f us | GPU busy %.1f us (%.1f %%) | >= 2 kernels in flight %.1f us | sum of kernel durations %.1f us | %d kernels'
          % (k, wall / 1e3, busy / 1e3, 100.0 * busy / wall, two / 1e3, ser / 1e3, len(seg)))
    gaps.sort(reverse=True)
    print('   idle gaps: total %.1f us in %d gaps; largest: %s' % (sum(g for g, _ in gaps) / 1e3, len(gaps), ', '.join('%.1f us at %.0f' % (g / 1e3, t / 1e3) for g, t in gaps[:8])))
    # phases: sweep (passA .. reconstruct_setup end), unwrap (first row kernel .. last phi_flush end), tail
    end_setup = max(int(r['End_Timestamp']) - t0 for r in seg if name(r) == 'reconstruct_setup_kernel')
    end_unw = max(int(r['End_Timestamp']) - t0 for r in seg if name(r) == 'phi_flush_kernel')
    print('   sweep phase %.1f us | unwrap phase %.1f us | tail to next passA %.1f us' % (end_setup / 1e3, (end_unw - end_setup) / 1e3, (wall - end_unw) / 1e3))
    if k == mid:
        d = collections.defaultdict(list)
        for r in seg: d[name(r)].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
        for nme, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
            print('      %-28s n=%3d total %8.1f us  median %7.2f' % (nme, len(v), sum(v), sorted(v)[len(v) // 2]))
PY
rm -rf $out/kt
cat $out/timeline_${n}_$dt.txt
