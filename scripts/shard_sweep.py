"""Kernel time of rank 0's 1/world shard of the bench frame over unit sizes and thresholds (tuning aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ray_tracing_fsharp_amd as rt
from ray_tracing_fsharp_amd import distributed as rtd
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
objs, cam, w, h = rt.sample_images.config3_final()
scene = rt.Scene.make(objs)
scene.tune(w, h, cam, seed=7)
rows, cols = 2 * h + 1, 2 * w + 1
first, stride, n = rtd.shard_rows(rows, 0, world)
local = torch.zeros((n, cols, 4), dtype=torch.int32, device="cuda:0")
for chunk in (0, 4, 8, 16):
    for refill in (0, 8, 12, 24):
        rt.set_launch_config(0, chunk); rt.set_schedule(0, refill)
        ts = []
        for _ in range(4):
            ts.append(rtd.render_shard_device(scene, cam, w, h, 2024, 0, first, stride, n, local, want_stats=True)["kernel_ms"])
        print(f"world {world} chunk {chunk:2d} refill {refill:2d}: {min(ts):.2f} ms", flush=True)
