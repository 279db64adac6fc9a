"""One line per render_kernel instantiation from csrc/resource_usage.txt (`make -C ray-tracing-fsharp_amd/csrc asm`)."""
import os
import re
import sys

path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "..", "ray-tracing-fsharp_amd", "csrc", "resource_usage.txt")
txt = open(path).read()
for b in re.split(r"remark: Function Name: ", txt)[1:]:
    name = b.split()[0]
    if "render_kernel" not in name:
        continue
    g = lambda k: re.search(k + r": (\d+)", b).group(1)
    m = re.search(r"render_kernelILb(\d)ELb(\d)ELi(\d+)ELi(\d)ELb(\d)", name)
    scratch = g(r"ScratchSize \[bytes/lane\]")
    print(f"LDS={m.group(1)} COUNT={m.group(2)} BLOCK={m.group(3):>4} MODE={m.group(4)} TEX={m.group(5)}  SGPR {g('TotalSGPRs'):>3} VGPR {g('VGPRs'):>3} "
          f"scratch {scratch:>4} B  SGPR-spill {g('SGPRs Spill'):>3}  VGPR-spill {g('VGPRs Spill'):>3}")
