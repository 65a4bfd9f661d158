O=$GRAFT_REPO_ROOT/gpurun_out/r03_bs; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_render.py tests/test_config5.py tests/test_level_sharded.py tests/test_cpp_demo.py -q -m gpu -x > $O/tests.log 2>&1; echo rc=$? >> $O/tests.log; tail -3 $O/tests.log
timeout -k 10 300 python3 tools/dbg/cull_campaign.py 20 2>&1 | tail -1
for W in 3 2 4; do export NMI_FRONT_WORKERS=$W; echo "workers $W: $(./examples/level_pipeline 200 | tail -3 | head -1 | grep -o '[0-9.]* levels/s')"; done
unset NMI_FRONT_WORKERS
cd /tmp && export TMPDIR=/tmp
rm -rf "$O/trace_cloud"
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$O/trace_cloud" -- python3 "$GRAFT_REPO_ROOT/bench.py" --config e2e --keyframes 20 > "$O/trace_cloud.log" 2>&1
python3 "$GRAFT_REPO_ROOT/tools/e2e_timeline.py" "$O/trace_cloud"
