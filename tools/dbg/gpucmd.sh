mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r03_x
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -x > gpurun_out/r03_x/tests.log 2>&1; echo rc=$? >> gpurun_out/r03_x/tests.log; tail -3 gpurun_out/r03_x/tests.log
for i in 1 2 3; do ./examples/relocalize_demo | grep "shim call site"; done > gpurun_out/r03_x/shim_rate.txt; cat gpurun_out/r03_x/shim_rate.txt
python3 tools/split_stamps.py 1 1 8 3 4 > gpurun_out/r03_x/split_stamps_pair_8x4.txt 2>&1; tail -9 gpurun_out/r03_x/split_stamps_pair_8x4.txt
