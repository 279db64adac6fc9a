cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/wr
for pk in 64 80 96 128; do
  for c in WRITE_SIZE FETCH_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d gpurun_out/wr/p${pk}_$c -o x -- python scripts/one_frame.py --park $pk > gpurun_out/wr/log_${pk}_$c.txt 2>&1
    python - gpurun_out/wr/p${pk}_$c $pk $c <<'PY'
import csv, sys, glob, re
f = glob.glob(sys.argv[1] + "/*counter_collection.csv")[0]
tot = 0.0
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "render_kernel" in k and not re.search(r"render_kernel<(true|false), true", k) and not re.search(r"\d+, 3,", k):
        tot += float(r["Counter_Value"])
print("park", sys.argv[2], sys.argv[3], round(tot / 1e6, 2), "GB (KB units)")
PY
  done
  python scripts/one_frame.py --park $pk --launches 3 2>/dev/null | grep -o "kernel_ms.: [0-9.]*" | tail -1
done
