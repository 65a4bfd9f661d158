O=$GRAFT_REPO_ROOT/gpurun_out/r03_ap; mkdir -p $O
timeout -k 10 400 python3 -m pytest tests/test_render.py tests/test_config5.py tests/test_level_sharded.py -q -m gpu -x > $O/tests.log 2>&1; echo rc=$? >> $O/tests.log; tail -15 $O/tests.log
cd /tmp && export TMPDIR=/tmp
for v in tiled scatter; do
  if [ $v = scatter ]; then export NMI_LEVEL_POINTS=scatter; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$v -- $GRAFT_REPO_ROOT/examples/level_pipeline 100 > $O/$v.log 2>&1
  echo "== $v"; python3 - $O/$v <<'PY'
import csv,sys,glob
f=glob.glob(f'{sys.argv[1]}/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:5]:
    print(r['Name'][:40], r['Calls'], r['AverageNs'])
PY
  tail -2 $O/$v.log
done
