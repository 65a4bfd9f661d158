mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r03_ac
python3 -m pytest tests -q -m gpu > gpurun_out/r03_ac/gputests.log 2>&1; echo rc=$? >> gpurun_out/r03_ac/gputests.log; tail -3 gpurun_out/r03_ac/gputests.log
for a in "" "--mesh" "--mesh 60x40"; do ./examples/level_pipeline 200 $a | tail -2 | head -1; done > gpurun_out/r03_ac/level_pipeline.txt 2>&1; cat gpurun_out/r03_ac/level_pipeline.txt
