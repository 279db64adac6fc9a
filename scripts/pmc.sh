#!/bin/bash
# PMC passes over one render launch; each counter group in its own rocprofv3 run (no trace domains mixed in).
# usage: scripts/pmc.sh <outdir> [one_frame.py args...]
set -u
OUT=$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
run() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o "$name" -- python scripts/one_frame.py $ARGS > "$OUT/$name.log" 2>&1 || echo "pass $name failed"; }
ARGS="$*"
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run sq2 SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VALU_FMA_F64
run sq3 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH
run ic SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH
run fetch FETCH_SIZE GRBM_GUI_ACTIVE
run write WRITE_SIZE
for d in sq1 sq2 sq3 ic fetch write; do f=$(ls $OUT/$d/*counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections, re
acc = collections.defaultdict(float)
for r in csv.DictReader(open(sys.argv[1])):
    # the frame's own launches: not the counting variant, and not the fused launch with the ray log (mode 3), which is rt_scene_tune's probe
    if "render_kernel" in r["Kernel_Name"] and not re.search(r"render_kernel<(true|false), true", r["Kernel_Name"]) and not re.search(r"render_kernel<(true|false), (true|false), \d+, 3,", r["Kernel_Name"]):
        acc[r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in acc.items():
    print(f"{k},{v:.0f}")
PY
done > "$OUT/summary.csv"
cat "$OUT/summary.csv"
