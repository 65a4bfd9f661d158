O=$GRAFT_REPO_ROOT/gpurun_out/r03_aq; mkdir -p $O
timeout -k 10 400 python3 -m pytest tests/test_render.py tests/test_config5.py tests/test_level_sharded.py -q -m gpu -x > $O/tests.log 2>&1; echo rc=$? >> $O/tests.log; tail -15 $O/tests.log
cd /tmp && export TMPDIR=/tmp
for d in 0 5; do
  export NMI_FRONT_DBG=$d
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/d$d -- $GRAFT_REPO_ROOT/examples/level_pipeline 100 > $O/d$d.log 2>&1
  echo "== dbg $d"; python3 - $O/d$d <<'PY'
import csv,sys,glob
f=glob.glob(f'{sys.argv[1]}/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:6]:
    print(r['Name'][:40], r['Calls'], r['AverageNs'])
PY
  tail -2 $O/d$d.log
done
