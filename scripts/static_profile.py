"""Static instruction profile of one kernel: VALU instructions per source line, from an assembly listing with .loc directives
(hipcc ... -gline-tables-only -S).  Inlined code is attributed to the line it was written on.
usage: python scripts/static_profile.py listing.s <kernel symbol> [--by-func ranges]"""
import collections
import re
import sys

path, sym = sys.argv[1], sys.argv[2]
files, cur, on = {}, None, False
per = collections.Counter()
kinds = collections.Counter()
for ln in open(path):
    m = re.match(r'\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', ln)
    if m:
        files[int(m.group(1))] = m.group(2)
        continue
    if ln.startswith(sym + ":"):
        on = True
        continue
    if not on:
        continue
    if ln.startswith(".Lfunc_end"):
        break
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", ln)
    if m:
        cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
        continue
    m = re.match(r"\s+(v_\w+|ds_\w+|s_\w+|global_\w+|scratch_\w+|buffer_\w+)", ln)
    if m:
        op = m.group(1)
        k = "valu" if op.startswith("v_") else "lds" if op.startswith("ds_") else "salu" if op.startswith("s_") else "vmem"
        kinds[k] += 1
        if k == "valu":
            per[cur] += 1
print(dict(kinds))
byfile = collections.Counter()
for (f, l), n in per.items():
    byfile[f] += n
print(dict(byfile))
for (f, l), n in sorted(per.items(), key=lambda kv: (kv[0][0], kv[0][1])):
    print(f"{f}:{l}\t{n}")
