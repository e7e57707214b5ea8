cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r01_v10 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/prof_r01_v10.log 2>&1
cd $GRAFT_REPO_ROOT && python3 bench.py > gpurun_out/bench_r01_v10.json 2> gpurun_out/bench_r01_v10.err; tail -c 600 gpurun_out/bench_r01_v10.json
