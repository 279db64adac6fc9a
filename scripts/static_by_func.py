"""Groups scripts/static_profile.py's per-line VALU counts by the function each line was written in (inlined code is
attributed to its source function).  usage: python scripts/static_by_func.py prof.txt"""
import collections
import os
import re
import sys

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ray-tracing-fsharp_amd", "csrc")
src = {f: open(os.path.join(root, f)).read().splitlines() for f in ("rt_device.h", "rt_render_kernel.h")}


def func_of(f, l):
    L = src[f]
    for i in range(l - 1, -1, -1):
        m = re.match(r"^\s{0,4}(?:template.*>\s*)?(?:RTD_INLINE|__device__|__global__|__host__)[^;]*?(\w+)\(", L[i])
        if m:
            return m.group(1)
    return "?"


acc = collections.Counter()
for ln in open(sys.argv[1]).read().splitlines()[2:]:
    key, n = ln.split("\t")
    f, l = key.rsplit(":", 1)
    f = f.split("/")[-1]
    acc[(f, func_of(f, int(l)) if f in src else "-")] += int(n)
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"{v:6d}  {k[0]}:{k[1]}")
