mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r03_fin
python3 -m pytest tests -q -m gpu > gpurun_out/r03_fin/gputests.log 2>&1; echo rc=$? >> gpurun_out/r03_fin/gputests.log; tail -3 gpurun_out/r03_fin/gputests.log
bash tools/collect_profiles.sh r03_final > gpurun_out/r03_fin/collect.log 2>&1; echo collect rc=$?
python3 bench.py --config e2e --map mesh > gpurun_out/r03_final/bench_e2e_mesh_4800.json 2>> gpurun_out/r03_final/bench.err
python3 bench.py --config e2e --map mesh --mesh-quads 300x200 > gpurun_out/r03_final/bench_e2e_mesh_120k.json 2>> gpurun_out/r03_final/bench.err
python3 tools/mesh_time.py > gpurun_out/r03_final/mesh_time.txt 2>&1
python3 tools/producer_time.py > gpurun_out/r03_final/producer_time.txt 2>&1
python3 tools/content_sensitivity.py > gpurun_out/r03_final/content_sensitivity.txt 2>&1
for a in "" "--mesh" "--mesh 60x40"; do ./examples/level_pipeline 200 $a | tail -2 | head -1; done > gpurun_out/r03_final/level_pipeline.txt 2>&1
for i in 1 2 3; do ./examples/relocalize_demo | grep "shim call site"; done > gpurun_out/r03_final/shim_rate.txt
cd /tmp && export TMPDIR=/tmp
for q in cloud 60x40 300x200; do if [ $q = cloud ]; then A=""; else A="--map mesh --mesh-quads $q"; fi; rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03_fin/trace_$q -- python3 $GRAFT_REPO_ROOT/bench.py --config e2e $A --keyframes 20 > $GRAFT_REPO_ROOT/gpurun_out/r03_fin/trace_$q.log 2>&1; python3 $GRAFT_REPO_ROOT/tools/e2e_timeline.py $GRAFT_REPO_ROOT/gpurun_out/r03_fin/trace_$q > $GRAFT_REPO_ROOT/gpurun_out/r03_final/e2e_timeline_$q.txt 2>&1; done
cat $GRAFT_REPO_ROOT/gpurun_out/r03_final/e2e_timeline_*.txt | grep "first start"
