"""Kernel time of rank 0's / the last rank's interleaved shard of the bench frame (predicts strong scaling), fused vs two-pass."""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ray_tracing_fsharp_amd as rt
from ray_tracing_fsharp_amd import distributed as rtd
objs, cam, w, h = rt.sample_images.config3_final()
scene = rt.Scene.make(objs)
if "--no-tune" not in sys.argv:
    print("tune:", scene.tune(w, h, cam, seed=2024))
rows, cols = 2*h+1, 2*w+1
base = None
ref = {}
for world in (1, 2, 4, 8):
    for passes, chunk in ((1, 16), (0, 0)):
        rt.set_launch_config(0, chunk); rt.set_passes(passes)
        ts = []
        for rank in (0, world - 1):
            first, stride, n = rtd.shard_rows(rows, rank, world)
            local = torch.zeros((n, cols, 4), dtype=torch.int32, device="cuda:0")
            for _ in range(3):
                st = rtd.render_shard_device(scene, cam, w, h, 2024, 0, first, stride, n, local, want_stats=True)
                ts.append(st["kernel_ms"])
            key = (world, rank)
            if key not in ref: ref[key] = local.clone()
            elif not torch.equal(ref[key], local): print("MISMATCH", world, rank, passes, chunk)
        t = statistics.median(ts)
        if base is None: base = t
        print(f"world {world} passes {passes} chunk {chunk:2d}: shard {t:8.2f} ms (max {max(ts):7.2f})  {base/max(ts):.2f}x of {world}", flush=True)
