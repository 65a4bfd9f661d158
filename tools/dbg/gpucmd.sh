O=$GRAFT_REPO_ROOT/gpurun_out/r03_ax; mkdir -p $O
timeout -k 10 500 python3 -m pytest tests/test_render.py -q -m gpu -x -k "common_planes or level" > $O/tests.log 2>&1; echo rc=$? >> $O/tests.log; tail -5 $O/tests.log
timeout -k 10 300 python3 tools/mesh_time.py > $O/mesh_time.txt 2>&1; cat $O/mesh_time.txt
for a in "" "--mesh" "--mesh 60x40"; do ./examples/level_pipeline 200 $a | tail -2 | head -1; done
