import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ray_tracing_fsharp_amd as rt
from ray_tracing_fsharp_amd import distributed as rtd
objs, cam, w, h = rt.sample_images.config3_final()
scene = rt.Scene.make(objs)
rows, cols = 2*h+1, 2*w+1
for world in (1, 4, 8):
    for chunk in (0,):
        rt.set_launch_config(0, chunk)
        first, stride, n = rtd.shard_rows(rows, 0, world)
        local = torch.zeros((n, cols, 4), dtype=torch.int32, device="cuda:0")
        st = rtd.render_shard_device(scene, cam, w, h, 2024, 0, first, stride, n, local, counters=True, want_stats=True)
        ss = (ctypes.c_uint64 * 16)(); rt.lib.rt_last_stage_stats(ss)
        life, span, waves = ss[6] / 100.0, ss[7] / 100.0, ss[8]   # microseconds
        print(f"world {world} chunk {chunk}: kernel {st['kernel_ms']:.2f} ms, waves {waves}, span {span/1e3:.2f} ms, mean wave lifetime {life/waves/1e3:.2f} ms "
              f"-> waves busy {life/waves/span*100:.1f} % of the span")
