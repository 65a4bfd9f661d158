O=$GRAFT_REPO_ROOT/gpurun_out/r03_bk; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for L in 500000 1000000 1600000; do
  export NMI_MESH_BOX_LANES=$L
  m=300x200
  rm -rf "$O/trace_$m"
  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$O/trace_$m" -- python3 "$GRAFT_REPO_ROOT/bench.py" --config e2e --keyframes 20 --map mesh --mesh-quads $m > "$O/trace_$m.log" 2>&1
  echo "box lanes $L"; python3 "$GRAFT_REPO_ROOT/tools/e2e_timeline.py" "$O/trace_$m" | grep "bin_kernel\|first start"
done
