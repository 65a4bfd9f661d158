O=$GRAFT_REPO_ROOT/gpurun_out/r03_bc; mkdir -p $O
timeout -k 10 500 python3 -m pytest tests/test_render.py tests/test_warp.py tests/test_config5.py -q -m gpu -x > $O/tests.log 2>&1; echo rc=$? >> $O/tests.log; tail -3 $O/tests.log
timeout -k 10 300 python3 tools/mesh_time.py 2>&1 | grep "queue on"
for a in "" "--mesh" "--mesh 60x40"; do ./examples/level_pipeline 200 $a | tail -2 | head -1; done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cloud -- $GRAFT_REPO_ROOT/examples/level_pipeline 100 > $O/cloud.log 2>&1
python3 - $O/cloud <<'PY'
import csv,sys,glob
f=glob.glob(f'{sys.argv[1]}/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:4]:
    print(r['Name'][:40], r['Calls'], r['AverageNs'])
PY
