#!/bin/bash
# Collects the round's measurement artefacts on the GPU box into gpurun_out/<tag>/:
#   bench.json (default bench line), bench_blocking.json, bench_stream.json, kernel trace summary, three PMC passes.
# Usage (on the GPU box, from the repo root):  bash tools/collect_profiles.sh r01_final
set -o pipefail
TAG=${1:-profile}
OUT=$PWD/gpurun_out/$TAG
ROOT=$PWD
mkdir -p "$OUT"
python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err" || exit 1                      # value = one blocking call per step, throughput beside it
python3 bench.py --throughput --no-cpu-baseline --no-call-site > "$OUT/bench_throughput.json" 2>> "$OUT/bench.err" || exit 1
python3 bench.py --throughput --streams 2 --no-cpu-baseline --no-call-site > "$OUT/bench_2streams.json" 2>> "$OUT/bench.err" || exit 1
python3 bench.py --config stream > "$OUT/bench_stream.json" 2>> "$OUT/bench.err" || exit 1
python3 bench.py --config e2e > "$OUT/bench_e2e.json" 2>> "$OUT/bench.err" || exit 1
python3 bench.py --config c3 --steps 300 --cpu-budget 6 > "$OUT/bench_c3.json" 2>> "$OUT/bench.err" || exit 1
python3 bench.py --config c4 --steps 300 --cpu-budget 6 > "$OUT/bench_c4.json" 2>> "$OUT/bench.err" || exit 1
python3 bench.py --gpus 2 --backend gloo --all-on-device0 --scaling strong --no-cpu-baseline --no-call-site > "$OUT/bench_gloo2_strong_rehearsal.json" 2>> "$OUT/bench.err" || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-call-site > "$OUT/trace.log" 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$ROOT/bench.py" --steps 20 --warmup 2 --no-cpu-baseline --no-call-site > "$OUT/pmc_fetch.log" 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$ROOT/bench.py" --steps 20 --warmup 2 --no-cpu-baseline --no-call-site > "$OUT/pmc_write.log" 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_c3" -- python3 "$ROOT/bench.py" --config c3 --steps 300 --no-cpu-baseline --no-call-site > "$OUT/trace_c3.log" 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_c4" -- python3 "$ROOT/bench.py" --config c4 --steps 300 --no-cpu-baseline --no-call-site > "$OUT/trace_c4.log" 2>&1 || exit 1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU --output-format csv -d "$OUT/pmc_lds" -- python3 "$ROOT/bench.py" --steps 20 --warmup 2 --no-cpu-baseline --no-call-site > "$OUT/pmc_lds.log" 2>&1 || exit 1
cd "$ROOT"
python3 tools/summarize_profiles.py "$OUT"
