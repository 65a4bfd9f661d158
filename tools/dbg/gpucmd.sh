O=$GRAFT_REPO_ROOT/gpurun_out/r03_bg; mkdir -p $O
timeout -k 10 500 python3 -m pytest tests/test_render.py -q -m gpu -x > $O/tests.log 2>&1; echo rc=$? >> $O/tests.log; tail -3 $O/tests.log
cd /tmp && export TMPDIR=/tmp
for v in old new; do
for m in "60 40" "300 200"; do
  export NMI_HIP_LIBRARY=$GRAFT_REPO_ROOT/build/ab/$v.so
  tag=${v}_$(echo $m | tr ' ' x)
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$tag -- python3 $GRAFT_REPO_ROOT/tools/mesh_profile.py $m 20 > $O/$tag.log 2>&1
  echo "$v mesh $m $(grep -h 'mesh_tile_kernel' $(find $O/$tag -name '*kernel_stats.csv') | awk -F, '{printf "%s %.1f  ", substr($1,7,18), $(NF-4)/1000}')"
done; done
