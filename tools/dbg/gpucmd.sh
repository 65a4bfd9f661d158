mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r03_r
cd /tmp && export TMPDIR=/tmp
for dbg in 0 1 2; do NMI_FRONT_DBG=$dbg rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03_r/trace_$dbg -- python3 $GRAFT_REPO_ROOT/bench.py --config e2e --keyframes 20 > $GRAFT_REPO_ROOT/gpurun_out/r03_r/trace_$dbg.log 2>&1; grep -h "front_kernel\|resolve" $GRAFT_REPO_ROOT/gpurun_out/r03_r/trace_$dbg/*/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-40,100-200; done
