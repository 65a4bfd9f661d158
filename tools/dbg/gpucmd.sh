O=$GRAFT_REPO_ROOT/gpurun_out/r03_bq; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_warp.py tests/test_render.py tests/test_config5.py -q -m gpu -x > $O/tests.log 2>&1; echo rc=$? >> $O/tests.log; tail -3 $O/tests.log
python3 tools/producer_time.py 2>/dev/null | grep -v amdgpu
for a in "" "--mesh" "--mesh 60x40"; do ./examples/level_pipeline 200 $a | tail -3 | head -1; done
cd /tmp && export TMPDIR=/tmp
for m in cloud 300x200; do
  rm -rf "$O/trace_$m"
  if [ $m = cloud ]; then args="--config e2e --keyframes 20"; else args="--config e2e --keyframes 20 --map mesh --mesh-quads $m"; fi
  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$O/trace_$m" -- python3 "$GRAFT_REPO_ROOT/bench.py" $args > "$O/trace_$m.log" 2>&1
  python3 "$GRAFT_REPO_ROOT/tools/e2e_timeline.py" "$O/trace_$m"
done
