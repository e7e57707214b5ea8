cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r01_final -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/prof_r01_final.log 2>&1
cd $GRAFT_REPO_ROOT && python3 bench.py > gpurun_out/bench_r01_final.json 2> gpurun_out/bench_r01_final.err; tail -c 600 gpurun_out/bench_r01_final.json
