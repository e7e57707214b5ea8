cd $GRAFT_REPO_ROOT
for e in "GPA_NO_RAW=1 GPA_NO_REORDER=1" "GPA_NO_RAW=1" "GPA_NO_REORDER=1" ""; do
  env $e timeout 600 python -m pytest tests/test_gpu_configs.py -x -q -k 'frames' 2>&1 | tail -1
  E="$e" python3 -c "
import json,os
d=json.load(open('gpurun_out/r02_accuracy.json'))
for k in ('frame_1000x1000_3x6_vs_oracle','frame_1080x1920_3x6_vs_oracle','frame_1280x1024_3x6_vs_oracle'):
    v=d[k]; print(os.environ['E'] or '(default)', k, {a:{x:round(b[x],9) for x in b if 'rms_px'==x or 'max_px'==x} for a,b in v.items() if isinstance(b,dict) and a in ('f32',)}, v['f32'].get('kidx_mismatch'))"
done
