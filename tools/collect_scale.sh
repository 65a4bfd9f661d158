#!/bin/bash
# One lease of an N-GPU node -> the whole scaling table: N in {1,2,4,8} x {c2 weak, c2 strong, c3 strong, c4 (= BASELINE
# configs[3], weak: 64 renders x 64 warps per rank), stream level-sharded, e2e level-sharded (cloud and mesh)} into
# profiles/<round>_scale/.  Usage: tools/collect_scale.sh [round tag, default r04] [largest N, default: devices visible]
set -u
cd "$(dirname "$0")/.."
TAG=${1:-r04}
MAXN=${2:-$(python3 -c 'import torch; print(torch.cuda.device_count())')}
OUT=profiles/${TAG}_scale
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() {  # name, args...
    local name=$1; shift
    echo "== $name: bench.py $*" >&2
    timeout -k 10 600 python3 bench.py "$@" > "$OUT/$name.json" 2> "$OUT/$name.err" || echo "   FAILED rc=$? (see $OUT/$name.err)" >&2
    [ -s "$OUT/$name.err" ] || rm -f "$OUT/$name.err"
}
for N in 1 2 4 8; do
    [ "$N" -le "$MAXN" ] || continue
    Q="--no-cpu-baseline --no-call-site"
    run c2_weak_n$N --gpus $N $Q
    run c4_weak_n$N --config c4 --gpus $N --steps 500 $Q
    if [ "$N" -gt 1 ]; then
        run c2_strong_n$N --gpus $N --scaling strong $Q
        run c3_strong_n$N --config c3 --gpus $N --scaling strong --steps 500 $Q
    fi
    run stream_level_n$N --config stream --shard level --gpus $N
    run e2e_level_n$N --config e2e --shard level --gpus $N
    run e2e_mesh_level_n$N --config e2e --map mesh --shard level --gpus $N
    run stream_replicas_n$N --config stream --gpus $N
done
python3 - "$OUT" <<'PY'
import glob, json, os, sys
rows = []
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.json"))):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        rows.append((os.path.basename(f)[:-5], d["n_gpus"], d.get("scaling"), d["value"], d["unit"], d.get("rccl")))
    except Exception as e:
        rows.append((os.path.basename(f)[:-5], "-", "-", f"unreadable: {e}", "", ""))
with open(os.path.join(sys.argv[1], "table.txt"), "w") as out:
    for r in rows:
        line = f"{r[0]:<26} n={r[1]} {r[2]} {r[3] if isinstance(r[3], str) else format(r[3], ',.0f')} {r[4]} rccl={r[5]}"
        print(line)
        out.write(line + "\n")
PY
