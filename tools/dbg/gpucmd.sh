O=$GRAFT_REPO_ROOT/gpurun_out/r03_br; mkdir -p $O
timeout -k 10 1000 python3 -m pytest tests -q -m gpu -x > $O/tests.log 2>&1; echo rc=$? >> $O/tests.log; tail -4 $O/tests.log
