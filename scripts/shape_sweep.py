"""The bench scene over image sizes and sample counts: Mray/s of one frame of the timed kernel (tuned tree), to find shapes where
the launch plan (unit sizes, fused or two passes) leaves the machine idle.
usage: python scripts/shape_sweep.py [--pixels 25 50 100 200 400 800] [--spp 4 16 64 256 1000]"""
import argparse
import dataclasses
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import ray_tracing_fsharp_amd as rt  # noqa: E402
from ray_tracing_fsharp_amd import distributed as rtd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--pixels", type=int, nargs="+", default=[25, 50, 100, 200, 400, 800])
ap.add_argument("--spp", type=int, nargs="+", default=[4, 16, 64, 256, 1000])
ap.add_argument("--depth", type=int, default=50)
ap.add_argument("--chunk", type=int, default=0, help="pixels per unit (0: the launch plan's own choice)")
ap.add_argument("--passes", type=int, default=0, help="1: fused, 2: two passes (0: the launch plan's own choice)")
a = ap.parse_args()
rt.set_launch_config(0, a.chunk)
rt.set_passes(a.passes)
print("pixels  image        " + "".join(f"{s:>10d}" for s in a.spp) + "   (Mray/s per spp)")
for px in a.pixels:
    line = ""
    for spp in a.spp:
        objs, cam, w, h = rt.sample_images.config3_final(spp=spp, depth=a.depth, pixels=px)
        scene = rt.Scene.make(objs)
        scene.tune(w, h, cam, seed=2024)
        rows, cols = 2 * h + 1, 2 * w + 1
        local = torch.zeros((rows, cols, 4), dtype=torch.int32, device="cuda:0")
        st = rtd.render_shard_device(scene, cam, w, h, 2024, 0, 0, 1, rows, local, counters=True, want_stats=True)
        best = min(rtd.render_shard_device(scene, cam, w, h, 2024, 0, 0, 1, rows, local, want_stats=True)["kernel_ms"] for _ in range(3))
        line += f"{st['rays'] / best / 1e3:10.0f}"
        if st["rays"] > 3e9:
            break
    print(f"{px:6d}  {2 * w + 1:5d}x{2 * h + 1:<5d}  {line}", flush=True)
