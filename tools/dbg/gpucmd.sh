mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r03_aj
timeout -k 10 400 python3 -m pytest tests/test_render.py -q -m gpu -k "soup or clipping or equal_depth" > gpurun_out/r03_aj/tests.log 2>&1; echo rc=$? >> gpurun_out/r03_aj/tests.log; tail -12 gpurun_out/r03_aj/tests.log
python3 tools/long_fuzz.py > gpurun_out/r03_aj/long_fuzz.txt 2>&1; cat gpurun_out/r03_aj/long_fuzz.txt
