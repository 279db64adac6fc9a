"""Interleaved A/B timing of render-kernel launch configurations in ONE process (cdna_hip_programming.md rule 24).
usage: python scripts/sweep.py [--rounds R] [--pixels P] [--spp S] block:chunk[:blocks_per_cu] ..."""
import argparse
import statistics
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import ray_tracing_fsharp_amd as rt  # noqa: E402
from ray_tracing_fsharp_amd import distributed as rtd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("configs", nargs="+")
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--pixels", type=int, default=800)
ap.add_argument("--spp", type=int, default=500)
ap.add_argument("--depth", type=int, default=50)
ap.add_argument("--counters", action="store_true")
args = ap.parse_args()

objs, cam, w, h = rt.sample_images.config3_final(spp=args.spp, depth=args.depth, pixels=args.pixels)
scene = rt.Scene.make(objs)
rows, cols = 2 * h + 1, 2 * w + 1
dev = torch.device("cuda", 0)
local = torch.zeros((rows, cols, 4), dtype=torch.int32, device=dev)
cfgs = [tuple(int(x) for x in c.split(":")) + (0,) * (5 - len(c.split(":"))) for c in args.configs]  # block:chunk:bpc:yield:refill
times = {c: [] for c in cfgs}
ref = None
for r in range(args.rounds + 1):
    for c in cfgs:
        rt.set_launch_config(*c[:3])
        rt.set_schedule(c[3], c[4])
        st = rtd.render_shard_device(scene, cam, w, h, 2024, 0, 0, 1, rows, local, counters=args.counters, want_stats=True)
        if r == 0:  # warm-up round doubles as a cross-config equality check
            cur = local.clone()
            if ref is None:
                ref = cur
            elif not torch.equal(ref, cur):
                print("MISMATCH between configs", c)
        else:
            times[c].append(st["kernel_ms"])
for c in cfgs:
    t = times[c]
    print(f"block={c[0]:5d} chunk={c[1]:3d} bpc={c[2]} yield={c[3]:2d} refill={c[4]:2d}  median {statistics.median(t):9.3f} ms  min {min(t):9.3f}  max {max(t):9.3f}")
