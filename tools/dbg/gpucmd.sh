O=$GRAFT_REPO_ROOT/gpurun_out/r03_bf; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_render.py tests/test_config5.py -q -m gpu -x -k "level or cloud or config5 or splat" > $O/tests.log 2>&1; echo rc=$? >> $O/tests.log; tail -3 $O/tests.log
cd /tmp && export TMPDIR=/tmp
for v in old new old new; do
  export NMI_HIP_LIBRARY=$GRAFT_REPO_ROOT/build/ab/$v.so
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$v -- $GRAFT_REPO_ROOT/examples/level_pipeline 100 > $O/$v.log 2>&1
  echo "$v $(grep -h front_kernel $(find $O/$v -name '*kernel_stats.csv') | awk -F, '{print $(NF-4)}' | tail -1) $(tail -2 $O/$v.log | head -1 | grep -o '[0-9.]* levels/s')"
  rm -rf $O/$v
done
