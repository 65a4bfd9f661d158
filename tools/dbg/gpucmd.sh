mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r03_s
timeout -k 10 400 python3 -m pytest tests/test_cpp_demo.py -q -m gpu > gpurun_out/r03_s/tests.log 2>&1; echo rc=$? >> gpurun_out/r03_s/tests.log; tail -5 gpurun_out/r03_s/tests.log
for a in "" "--mesh" "--mesh 60x40"; do ./examples/level_pipeline 200 $a | tail -3; done > gpurun_out/r03_s/level_pipeline.txt 2>&1; cat gpurun_out/r03_s/level_pipeline.txt
