#!/bin/bash
# Counters under the default-mode level's producer kernels, on the GPU box from the repo root:
#   bash tools/collect_profiles_producers.sh r04_b
# For the three maps of bench.py --config e2e (3 M-point cloud, 4,800- and 120 k-triangle textured meshes): a kernel trace
# (timeline of one level) and three rocprofv3 --pmc passes of their own (SQ counters; FETCH_SIZE; WRITE_SIZE -- the TCC block has
# 4 slots, FETCH_SIZE takes 3 and WRITE_SIZE 2), then tools/summarize_producers.py -> pmc_producers.json.
set -o pipefail
TAG=${1:-profile}
OUT=$PWD/gpurun_out/$TAG
ROOT=$PWD
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_WAIT_ANY"
for m in cloud 60x40 300x200; do
  if [ $m = cloud ]; then args="--config e2e"; else args="--config e2e --map mesh --mesh-quads $m"; fi
  rm -rf "$OUT/trace_$m" "$OUT/pmc_sq_$m" "$OUT/pmc_fetch_$m" "$OUT/pmc_write_$m"
  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$OUT/trace_$m" -- python3 "$ROOT/bench.py" $args --keyframes 20 > "$OUT/trace_$m.log" 2>&1 || exit 1
  python3 "$ROOT/tools/e2e_timeline.py" "$OUT/trace_$m" > "$OUT/e2e_timeline_$m.txt" || exit 1
  rocprofv3 --pmc $SQ --output-format csv -d "$OUT/pmc_sq_$m" -- python3 "$ROOT/bench.py" $args --keyframes 6 > "$OUT/pmc_sq_$m.log" 2>&1 || exit 1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch_$m" -- python3 "$ROOT/bench.py" $args --keyframes 6 > "$OUT/pmc_fetch_$m.log" 2>&1 || exit 1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write_$m" -- python3 "$ROOT/bench.py" $args --keyframes 6 > "$OUT/pmc_write_$m.log" 2>&1 || exit 1
  echo "== $m done" >&2
done
cd "$ROOT"
python3 tools/summarize_producers.py "$OUT"
