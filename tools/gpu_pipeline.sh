#!/bin/bash
# usage (GPU box): tools/gpu_pipeline.sh [size] [dtype]   -> gpurun_out/pipeline/{pipeline_<n>_<dt>.json, kernel_stats_pipeline_<n>_<dt>.csv}
# the image -> k-vectors -> u -> undistortion -> properties leg of bench.py alone, then the same under rocprofv3 --kernel-trace --stats
ulimit -c 0
ROOT=$GRAFT_REPO_ROOT
n=${1:-4096}; dt=${2:-f32}
out=$ROOT/gpurun_out/pipeline; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py --only-pipeline --size $n --dtype $dt > $out/pipeline_${n}_$dt.json 2> $out/pipeline_${n}_$dt.err
python3 - <<PY
import json
d = json.load(open('$out/pipeline_${n}_$dt.json'))['pipeline_end_to_end']
print(d['value'], d['unit'], d['ms_per_image'], 'ms', d['stage_ms'], 'k error', d['found_kvectors_max_error_cycles_per_px'])
for st, rows in d['kernels'].items():
    print(st)
    for k, r in rows.items():
        print('   %-28s x%-3d %8.3f ms %s' % (k, r['launches'], r['total_ms'], ('%.2f of HBM on %.3f GB' % (r['frac_of_hbm_peak'], r['compulsory_GB'])) if 'frac_of_hbm_peak' in r else ''))
PY
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ks -- python3 $ROOT/bench.py --only-pipeline --size $n --dtype $dt > $out/ks.log 2>&1
f=$(ls $out/ks/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp $f $out/kernel_stats_pipeline_${n}_$dt.csv
rm -rf $out/ks
