"""One launch of the render kernel on the bench workload (config 3), for rocprofv3 --pmc passes.
usage: python scripts/one_frame.py [--block B] [--chunk C] [--pixels P] [--spp S] [--launches L]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import ray_tracing_fsharp_amd as rt  # noqa: E402
from ray_tracing_fsharp_amd import distributed as rtd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--block", type=int, default=0)
ap.add_argument("--chunk", type=int, default=0)
ap.add_argument("--pixels", type=int, default=800)
ap.add_argument("--spp", type=int, default=500)
ap.add_argument("--depth", type=int, default=50)
ap.add_argument("--launches", type=int, default=1)
ap.add_argument("--counters", action="store_true")
ap.add_argument("--park", type=int, default=0, help="park pool entries per wave; -1 = never park")
ap.add_argument("--passes", type=int, default=0)
ap.add_argument("--yield-lanes", type=int, default=0)
ap.add_argument("--refill-lanes", type=int, default=0)
ap.add_argument("--no-tune", action="store_true", help="skip rt_scene_tune")
ap.add_argument("--grid", type=int, default=11, help="the scene's recipe over a larger grid (sample_images.randomSpheres): 16 -> 1026 spheres, 26 -> 2705")
ap.add_argument("--stage-stats", action="store_true", help="print rt_last_stage_stats of a timed launch too (a -DRTD_STAGE_CLOCKS build fills the cycle sums)")
a = ap.parse_args()
rt.set_launch_config(a.block, a.chunk)
rt.set_park(a.park)
rt.set_passes(a.passes)
rt.set_schedule(a.yield_lanes, a.refill_lanes)
objs, cam, w, h = rt.sample_images.config3_final(spp=a.spp, depth=a.depth, pixels=a.pixels)
if a.grid != 11:
    import dataclasses
    objs, cam, w, h = rt.sample_images.randomSpheres(2024, a.spp, a.pixels, grid=a.grid)
    cam = dataclasses.replace(cam, BounceDepth=a.depth)
scene = rt.Scene.make(objs)
if not a.no_tune:
    scene.tune(w, h, cam, seed=2024)
rows, cols = 2 * h + 1, 2 * w + 1
local = torch.zeros((rows, cols, 4), dtype=torch.int32, device="cuda:0")
for _ in range(a.launches):
    st = rtd.render_shard_device(scene, cam, w, h, 2024, 0, 0, 1, rows, local, counters=a.counters, want_stats=True)
    print({k: st[k] for k in ("kernel_ms", "rays", "aabb_tests", "prim_tests", "samples")}, flush=True)
    if a.counters or a.stage_stats:
        import ctypes
        ss = (ctypes.c_uint64 * 16)()
        rt.lib.rt_last_stage_stats(ss)
        names = ("refill_stages", "node_trips", "leaf_stages", "shade_stages", "lanes_refilled", "lanes_shaded", "wave_ticks", "span_ticks", "waves", "slow_stages", "slow_lanes", "parked_lanes", "cyc_refill", "cyc_slow", "cyc_walk", "cyc_shade")
        print(dict(zip(names, list(ss))), flush=True)
