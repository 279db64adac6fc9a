"""How evenly a shard's waves finish: counting launch of rank 0's interleaved shard of the bench frame, sum of wave lifetimes / (waves x span).
usage: python scripts/shard_balance.py [world ...]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import ray_tracing_fsharp_amd as rt  # noqa: E402
from ray_tracing_fsharp_amd import distributed as rtd  # noqa: E402

objs, cam, w, h = rt.sample_images.config3_final()
scene = rt.Scene.make(objs)
scene.tune(w, h, cam, seed=2024)
rows, cols = 2 * h + 1, 2 * w + 1
for world in [int(x) for x in sys.argv[1:]] or [1, 2, 4, 8]:
    first, stride, n = rtd.shard_rows(rows, 0, world)
    local = torch.zeros((n, cols, 4), dtype=torch.int32, device="cuda:0")
    st = rtd.render_shard_device(scene, cam, w, h, 2024, 0, first, stride, n, local, counters=True, want_stats=True)
    ss = (ctypes.c_uint64 * 16)()
    rt.lib.rt_last_stage_stats(ss)
    wave_ticks, span, waves = ss[6], ss[7], ss[8]
    t = rtd.render_shard_device(scene, cam, w, h, 2024, 0, first, stride, n, local, want_stats=True)
    print({"world": world, "counting_kernel_ms": round(st["kernel_ms"], 2), "timed_kernel_ms": round(t["kernel_ms"], 2), "waves": waves, "span_ms": round(span / 1e5, 2),
           "mean_wave_lifetime_ms": round(wave_ticks / max(1, waves) / 1e5, 2), "balance": round(wave_ticks / max(1, waves * span), 3)}, flush=True)
