for v in old new old new; do
  export NMI_HIP_LIBRARY=$GRAFT_REPO_ROOT/build/ab/$v.so
  echo "== $v"; python3 tools/producer_time.py 2>&1 | grep "point render"
done
