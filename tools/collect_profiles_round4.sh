#!/bin/bash
# Round 4's additional artefacts (on the GPU box, from the repo root):  bash tools/collect_profiles_round4.sh r04_final
# small / mid-size grids per kernel selection, the pixel-range kernel's stamps, odd widths, mesh renderer times, the three level
# pipelines, the unchanged call site from C++, e2e bench lines (with their per-kernel roofline blocks), the long fuzz campaign.
set -o pipefail
TAG=${1:-profile}
OUT=$PWD/gpurun_out/$TAG
ROOT=$PWD
mkdir -p "$OUT"
python3 tools/small_grid_time.py 2>/dev/null | grep -v amdgpu.ids > "$OUT/small_grid_time.txt" || exit 1
python3 tools/pix_stamps.py 9 9 3 2>/dev/null | grep -v amdgpu.ids > "$OUT/pix_stamps_9x9_3.txt" || exit 1
python3 tools/pix_stamps.py 27 4 2 2>/dev/null | grep -v amdgpu.ids > "$OUT/pix_stamps_27x4_2.txt" || exit 1
python3 tools/grid_stamps.py 9 9 2>/dev/null | grep -v amdgpu.ids > "$OUT/grid_stamps_9x9.txt" || exit 1
python3 tools/odd_width_time.py 2>/dev/null | grep -v amdgpu.ids > "$OUT/odd_width_time.txt" || exit 1
python3 tools/mesh_time.py 2>/dev/null | grep -v amdgpu.ids > "$OUT/mesh_time.txt" || exit 1
{ ./examples/level_pipeline 200; ./examples/level_pipeline 200 --mesh; ./examples/level_pipeline 200 --mesh 60x40; } 2>&1 | grep "levels/s\|PIPELINE" > "$OUT/level_pipeline.txt" || exit 1
./examples/relocalize_demo 2>&1 | grep -i "evals\|SHIM\|find_max" > "$OUT/shim_rate.txt" || exit 1
python3 bench.py --config e2e > "$OUT/bench_e2e.json" 2>> "$OUT/bench.err" || exit 1
python3 bench.py --config e2e --map mesh --mesh-quads 300x200 > "$OUT/bench_e2e_mesh_120k.json" 2>> "$OUT/bench.err" || exit 1
python3 bench.py --config e2e --map mesh --mesh-quads 60x40 > "$OUT/bench_e2e_mesh_4800.json" 2>> "$OUT/bench.err" || exit 1
python3 tools/long_fuzz.py > "$OUT/long_fuzz.txt" 2>&1 || exit 1
tail -3 "$OUT/long_fuzz.txt"; cat "$OUT/small_grid_time.txt" "$OUT/level_pipeline.txt" "$OUT/shim_rate.txt" "$OUT/mesh_time.txt"
