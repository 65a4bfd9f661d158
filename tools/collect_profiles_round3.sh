#!/bin/bash
# Round 3's additional artefacts (on the GPU box, from the repo root):  bash tools/collect_profiles_round3.sh r03_final
# mesh renderer times, the three level pipelines (cloud, 120 k and 4,800 triangles), their per-kernel timelines, the e2e bench
# lines with a mesh as the map, the unchanged call site from C++, the long fuzz campaign's summary.
set -o pipefail
TAG=${1:-profile}
OUT=$PWD/gpurun_out/$TAG
ROOT=$PWD
mkdir -p "$OUT"
python3 tools/mesh_time.py 2>/dev/null | grep -v amdgpu.ids > "$OUT/mesh_time.txt" || exit 1
python3 tools/producer_time.py 2>/dev/null | grep -v amdgpu.ids > "$OUT/producer_time.txt" || exit 1
{ ./examples/level_pipeline 200; ./examples/level_pipeline 200 --mesh; ./examples/level_pipeline 200 --mesh 60x40; } 2>&1 | grep "levels/s\|PIPELINE" > "$OUT/level_pipeline.txt" || exit 1
./examples/relocalize_demo 2>&1 | grep -i "evals\|SHIM" > "$OUT/shim_rate.txt" || exit 1
python3 bench.py --config e2e --map mesh --mesh-quads 300x200 > "$OUT/bench_e2e_mesh_120k.json" 2>> "$OUT/bench.err" || exit 1
python3 bench.py --config e2e --map mesh --mesh-quads 60x40 > "$OUT/bench_e2e_mesh_4800.json" 2>> "$OUT/bench.err" || exit 1
cd /tmp && export TMPDIR=/tmp
for m in cloud 300x200 60x40; do
  rm -rf "$OUT/trace_$m"
  if [ $m = cloud ]; then args="--config e2e --keyframes 20"; else args="--config e2e --keyframes 20 --map mesh --mesh-quads $m"; fi
  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$OUT/trace_$m" -- python3 "$ROOT/bench.py" $args > "$OUT/trace_$m.log" 2>&1 || exit 1
  python3 "$ROOT/tools/e2e_timeline.py" "$OUT/trace_$m" > "$OUT/e2e_timeline_$m.txt" || exit 1
done
cd "$ROOT"
python3 tools/long_fuzz.py > "$OUT/long_fuzz.txt" 2>&1 || exit 1
tail -3 "$OUT/long_fuzz.txt"; cat "$OUT/mesh_time.txt" "$OUT/level_pipeline.txt" "$OUT/shim_rate.txt"; cat "$OUT"/e2e_timeline_*.txt
