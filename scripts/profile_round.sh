#!/bin/bash
# Everything under profiles/<round>/ from one GPU call: the VALU issue-rate microbenchmark, the bench line with the
# rocprofv3 kernel-trace summary of the same command, and the PMC passes (each counter group in a run of its own).
# usage (on the GPU box): bash scripts/profile_round.sh r2 <git-hash>
set -u
ROUND=${1:-r2}; HASH=${2:-unknown}
OUT=gpurun_out/$ROUND; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rates scripts/ubench/valu_rates.hip && /tmp/valu_rates > $OUT/valu_rates.txt 2>&1
python bench.py --steps 10 --warmup 2 > $OUT/bench_n1_$HASH.json 2> $OUT/bench_n1.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python bench.py --steps 10 --warmup 2 --cpu-seconds 0 --no-untuned-leg > $OUT/bench_traced.json 2> $OUT/trace.err
f=$(ls $OUT/trace/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && { echo "# git $HASH: rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 2 --cpu-seconds 0 --no-untuned-leg"; cat "$f"; } > $OUT/bench_n1_${HASH}_kernel_stats.csv
bash scripts/pmc.sh $OUT/pmc > $OUT/pmc.log 2>&1
{ echo "# git $HASH: scripts/pmc.sh (separate rocprofv3 --pmc passes over one launch of the bench frame, scripts/one_frame.py); sums over the frame's own render_kernel dispatches (pass A + pass B; not the probe of rt_scene_tune)"; cat $OUT/pmc/summary.csv; } > $OUT/pmc_bench_config_summary.csv
tail -2 $OUT/valu_rates.txt; cat $OUT/bench_n1_$HASH.json; cat $OUT/bench_n1_${HASH}_kernel_stats.csv | head -8; cat $OUT/pmc_bench_config_summary.csv
