mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r03_ai
timeout -k 10 400 python3 -m pytest tests/test_render.py tests/test_level_sharded.py tests/test_config5.py -q -m gpu > gpurun_out/r03_ai/tests.log 2>&1; echo rc=$? >> gpurun_out/r03_ai/tests.log; tail -3 gpurun_out/r03_ai/tests.log
timeout -k 10 200 python3 tools/mesh_time.py > gpurun_out/r03_ai/mesh_time.txt 2>&1; cat gpurun_out/r03_ai/mesh_time.txt
