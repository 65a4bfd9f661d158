set -o pipefail
# Re-takes the headline profile (bench line, rocprofv3 --kernel-trace --stats, the three --pmc passes, grid stamps) into gpurun_out/r04_final
# after a late change of the kernel sources, so that bench.py's static stamps (roofline.static) are fresh.  On the GPU box, repo root.
cd /root/repo
OUT=$PWD/gpurun_out/r04_final
ROOT=$PWD
mkdir -p "$OUT"
python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err" || exit 1
cd /tmp && export TMPDIR=/tmp
rm -rf "$OUT/trace" "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_lds"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-call-site > "$OUT/trace.log" 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$ROOT/bench.py" --steps 20 --warmup 2 --no-cpu-baseline --no-call-site > "$OUT/pmc_fetch.log" 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$ROOT/bench.py" --steps 20 --warmup 2 --no-cpu-baseline --no-call-site > "$OUT/pmc_write.log" 2>&1 || exit 1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU --output-format csv -d "$OUT/pmc_lds" -- python3 "$ROOT/bench.py" --steps 20 --warmup 2 --no-cpu-baseline --no-call-site > "$OUT/pmc_lds.log" 2>&1 || exit 1
cd "$ROOT"
python3 tools/summarize_profiles.py "$OUT" | tail -12
python3 tools/grid_stamps.py 9 9 2>/dev/null | grep -v amdgpu.ids > "$OUT/grid_stamps_9x9.txt"
