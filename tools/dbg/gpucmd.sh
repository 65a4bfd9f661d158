O=$GRAFT_REPO_ROOT/gpurun_out/r03_bm; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_render.py tests/test_config5.py tests/test_level_sharded.py -q -m gpu -x -k "level or cull or config5" > $O/tests.log 2>&1; echo rc=$? >> $O/tests.log; tail -3 $O/tests.log
timeout -k 10 300 python3 tools/dbg/cull_campaign.py 20 2>&1 | tail -1
for i in 1 2; do ./examples/level_pipeline 300 | tail -2 | head -1; done
