O=$GRAFT_REPO_ROOT/gpurun_out/r03_bh; mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests -q -m gpu -x > $O/tests.log 2>&1; echo rc=$? >> $O/tests.log; tail -5 $O/tests.log
timeout -k 10 120 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
