"""Static instruction counts between the `; MARK <name>` comments of a kernel listing (a scratch build of rt_render_kernel.h with
asm volatile("; MARK ...") at the stage boundaries): VALU / SALU / LDS / VMEM per stage.  usage: python scripts/stage_static_counts.py hot.s"""
import sys

L = open(sys.argv[1]).read().split("\n")
marks = [(i, l.split("MARK")[1].strip()) for i, l in enumerate(L) if "MARK" in l] + [(len(L), "end")]


def count(a, b):
    v = s = m = d = 0
    for l in L[a:b]:
        t = l.strip()
        if not t or t[0] in ";." or t.endswith(":"):
            continue
        op = t.split()[0]
        v += op.startswith("v_"); s += op.startswith("s_"); d += op.startswith("ds_")
        m += op.startswith(("global_", "scratch_", "buffer_", "flat_"))
    return v, s, d, m


prev = (0, "start")
for i, name in marks:
    c = count(prev[0], i)
    print(f"{prev[1]:>15} -> {name:<15} VALU {c[0]:4d} SALU {c[1]:4d} LDS {c[2]:3d} VMEM {c[3]:3d}")
    prev = (i, name)
