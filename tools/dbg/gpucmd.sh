O=$GRAFT_REPO_ROOT/gpurun_out/r03_al; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in a b a2 b2; do
  case $v in a*) unset NMI_LEVEL_NO_BOUND;; b*) export NMI_LEVEL_NO_BOUND=1;; esac
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$v -- $GRAFT_REPO_ROOT/examples/level_pipeline 100 > $O/$v.log 2>&1
  echo "== $v"; grep -h "front\|resolve" $(find $O/$v -name "*kernel_stats.csv") | cut -d, -f1-4
done
