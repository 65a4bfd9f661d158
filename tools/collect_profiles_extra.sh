#!/bin/bash
# Second half of the round's artefacts (on the GPU box, from the repo root):  bash tools/collect_profiles_extra.sh r02_final
# content sensitivity, producers, small grids, the unchanged call site from C++, the level pipeline, the e2e timeline.
set -o pipefail
TAG=${1:-profile}
OUT=$PWD/gpurun_out/$TAG
ROOT=$PWD
mkdir -p "$OUT"
python3 tools/content_sensitivity.py > "$OUT/content_sensitivity.txt" 2>/dev/null || exit 1
python3 tools/producer_time.py > "$OUT/producer_time.txt" 2>/dev/null || exit 1
python3 tools/small_grid_time.py > "$OUT/small_grid_time.txt" 2>/dev/null || exit 1
python3 tools/small_grid_content.py 2>/dev/null | grep levels= > "$OUT/small_grid_content.txt" || exit 1
python3 tools/bg_off_time.py 2>/dev/null | grep use_bg > "$OUT/bg_off_time.txt" || exit 1
python3 tools/mesh_time.py > "$OUT/mesh_time.txt" 2>/dev/null || exit 1
./examples/relocalize_demo 2>&1 | grep -i "evals\|SHIM" > "$OUT/shim_rate.txt" || exit 1
python3 tools/cloud_order_time.py 2>/dev/null | grep -v amdgpu.ids > "$OUT/cloud_order_time.txt" || exit 1
[ -x tools/ubench/launch_latency ] || hipcc --offload-arch=gfx950 -O2 -o tools/ubench/launch_latency tools/ubench/launch_latency.hip 2>/dev/null
./tools/ubench/launch_latency > "$OUT/launch_latency.txt" || exit 1
./examples/level_pipeline > "$OUT/level_pipeline.txt" 2>&1 || exit 1
cd /tmp && export TMPDIR=/tmp
rm -rf "$OUT/e2e_trace"
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$OUT/e2e_trace" -- python3 "$ROOT/bench.py" --config e2e --keyframes 20 > "$OUT/e2e_trace.log" 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/few_trace" -o few -- python3 "$ROOT/tools/few_levels_time.py" > "$OUT/few_levels_time.txt" 2>&1 || exit 1
cd "$ROOT"
python3 tools/e2e_timeline.py "$OUT/e2e_trace" > "$OUT/e2e_timeline.txt" || exit 1
