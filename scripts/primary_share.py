"""What share of a frame's box tests and leaf tests belongs to the PRIMARY rays (the camera rays): the counting variant at bounce
depth 0 (every sample traces exactly its camera ray; with adaptive sampling switched off by comparing at equal sample counts)."""
import dataclasses, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ray_tracing_fsharp_amd as rt
from ray_tracing_fsharp_amd import distributed as rtd, sample_images as si
objs, cam, w, h = si.config3_final(spp=500, depth=50)
scene = rt.Scene.make(objs); scene.tune(w, h, cam, seed=2024 ^ 0x5EED)
rows, cols = 2 * h + 1, 2 * w + 1
local = torch.zeros((rows, cols, 4), dtype=torch.int32, device="cuda:0")
full = rtd.render_shard_device(scene, cam, w, h, 2024, 0, 0, 1, rows, local, counters=True, want_stats=True)
cam0 = dataclasses.replace(cam, BounceDepth=0)
prim = rtd.render_shard_device(scene, cam0, w, h, 2024, 0, 0, 1, rows, local, counters=True, want_stats=True)
for k in ("rays", "aabb_tests", "prim_tests", "samples"):
    print(k, full[k], prim[k], round(prim[k] / full[k], 4))
print("box tests per ray: all", full["aabb_tests"] / full["rays"], "primary", prim["aabb_tests"] / prim["rays"])
