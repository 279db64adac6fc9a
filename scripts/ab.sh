#!/bin/bash
# A/B of kernel builds on the GPU box: for every library given, the bench frame (config 3) timed, and one counting launch with the
# stage statistics.  usage: bash scripts/ab.sh <outdir> <lib.so|default> [...]   (extra bench flags via AB_FLAGS)
set -u
OUT=$1; shift
mkdir -p "$OUT"
for lib in "$@"; do
    name=$(basename "$lib" .so)
    if [ "$lib" = default ]; then unset RTFS_LIB; else export RTFS_LIB="$PWD/$lib"; fi
    python bench.py --steps 10 --warmup 2 --cpu-seconds 0 --no-untuned-leg ${AB_FLAGS:-} > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err" || echo "bench $name failed"
    python scripts/one_frame.py --stage-stats > "$OUT/stages_$name.txt" 2>&1 || echo "stages $name failed"
    python - "$OUT/bench_$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "ms_per_step", d["ms_per_step"], "kernel_ms", d["roofline"]["kernel_ms"], "Mray/s", d["value"], flush=True)
except Exception as e:
    print(sys.argv[2], "unreadable:", e, flush=True)
PY
done
