mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r03_ab
timeout -k 10 400 python3 -m pytest tests/test_render.py tests/test_config5.py tests/test_level_sharded.py -q -m gpu > gpurun_out/r03_ab/tests.log 2>&1; echo rc=$? >> gpurun_out/r03_ab/tests.log; tail -3 gpurun_out/r03_ab/tests.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03_ab/trace_cloud -- python3 $GRAFT_REPO_ROOT/bench.py --config e2e --keyframes 20 > $GRAFT_REPO_ROOT/gpurun_out/r03_ab/trace_cloud.log 2>&1; python3 $GRAFT_REPO_ROOT/tools/e2e_timeline.py $GRAFT_REPO_ROOT/gpurun_out/r03_ab/trace_cloud > $GRAFT_REPO_ROOT/gpurun_out/r03_ab/e2e_timeline_cloud.txt 2>&1; cat $GRAFT_REPO_ROOT/gpurun_out/r03_ab/e2e_timeline_cloud.txt
