O=$GRAFT_REPO_ROOT/gpurun_out/r03_ao; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for d in 5 13 21 37; do
  export NMI_FRONT_DBG=$d
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/d$d -- $GRAFT_REPO_ROOT/examples/level_pipeline 40 > $O/d$d.log 2>&1
  echo "dbg=$d $(grep -h front_kernel $(find $O/d$d -name '*kernel_stats.csv') | awk -F, '{print $(NF-6), $(NF-5), $(NF-4)}')"
done
