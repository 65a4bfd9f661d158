mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r03_m
timeout -k 10 300 python3 -m pytest tests/test_render.py tests/test_level_sharded.py -q -m gpu > gpurun_out/r03_m/mesh_tests.log 2>&1; echo rc=$? >> gpurun_out/r03_m/mesh_tests.log; tail -4 gpurun_out/r03_m/mesh_tests.log
timeout -k 10 200 python3 tools/mesh_time.py > gpurun_out/r03_m/mesh_time.txt 2>&1; cat gpurun_out/r03_m/mesh_time.txt
cd /tmp && export TMPDIR=/tmp
for dbg in 0 1 3; do for c in "60 40" "300 200"; do set -- $c; NMI_MESH_DBG=$dbg rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03_m/prof_${1}_$dbg -- python3 $GRAFT_REPO_ROOT/tools/mesh_profile.py $1 $2 > $GRAFT_REPO_ROOT/gpurun_out/r03_m/prof_${1}_$dbg.log 2>&1; done; done
