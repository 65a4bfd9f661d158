mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r03_y
python3 tools/small_grid_time.py > gpurun_out/r03_y/small_grid_time.txt 2>&1; cat gpurun_out/r03_y/small_grid_time.txt | grep -v "^ *[13]x1\|^ *9x1" 
