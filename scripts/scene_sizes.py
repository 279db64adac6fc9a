"""Per-ray time against scene size: the final scene's recipe over larger grids (sample_images.randomSpheres(grid=...)), same camera,
same image, reduced spp.  Says where a scene stops fitting the LDS and what the global-memory variant of the kernel costs.
usage: python scripts/scene_sizes.py [--grids 11 12 16 26] [--spp 100] [--pixels 400]"""
import argparse
import dataclasses
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import ray_tracing_fsharp_amd as rt  # noqa: E402
from ray_tracing_fsharp_amd import distributed as rtd  # noqa: E402
from ray_tracing_fsharp_amd import sample_images as si  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--grids", type=int, nargs="+", default=[11, 12, 16, 26])
ap.add_argument("--spp", type=int, default=100)
ap.add_argument("--pixels", type=int, default=400)
ap.add_argument("--tune", action="store_true")
a = ap.parse_args()
for grid in a.grids:
    objs, cam, w, h = si.randomSpheres(2024, a.spp, a.pixels, grid=grid)
    cam = dataclasses.replace(cam, BounceDepth=50)
    scene = rt.Scene.make(objs)
    if a.tune:
        scene.tune(w, h, cam, seed=7)
    info = scene.info()
    rows, cols = 2 * h + 1, 2 * w + 1
    local = torch.zeros((rows, cols, 4), dtype=torch.int32, device="cuda:0")
    st = rtd.render_shard_device(scene, cam, w, h, 2024, 0, 0, 1, rows, local, counters=True, want_stats=True)
    best = None
    for _ in range(3):
        t = rtd.render_shard_device(scene, cam, w, h, 2024, 0, 0, 1, rows, local, want_stats=True)
        best = t["kernel_ms"] if best is None else min(best, t["kernel_ms"])
    print(json.dumps({"grid": grid, "spheres": info["n_bounded"], "walk_tree_nodes": info["walk_tree_nodes"], "lds_resident": info["lds_resident"],
                      "leaf_box_implied": info["leaf_box_implied"], "rays": st["rays"], "box_tests_per_ray": round(st["aabb_tests"] / st["rays"], 2),
                      "kernel_ms": round(best, 3), "ns_per_ray": round(best * 1e6 / st["rays"], 4), "Mray_per_s": round(st["rays"] / best / 1e3, 1)}), flush=True)
