import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ray_tracing_fsharp_amd as rt
from ray_tracing_fsharp_amd import distributed as rtd
objs, cam, w, h = rt.sample_images.config3_final()
scene = rt.Scene.make(objs)
rows, cols = 2*h+1, 2*w+1
names = ("refill_stages", "node_trips", "leaf_stages", "shade_stages", "lanes_refilled", "lanes_shaded")
local = torch.zeros((rows, cols, 4), dtype=torch.int32, device="cuda:0")
for passes, chunk in ((1, 16), (2, 16), (2, 4), (2, 2)):
    rt.set_launch_config(0, chunk); rt.set_passes(passes)
    st = rtd.render_shard_device(scene, cam, w, h, 2024, 0, 0, 1, rows, local, counters=True, want_stats=True)
    ss = (ctypes.c_uint64 * 16)(); rt.lib.rt_last_stage_stats(ss); d = dict(zip(names, list(ss)))
    print(f"passes {passes} chunk {chunk}: {st['kernel_ms']:.1f} ms  node_trips {d['node_trips']/1e9:.3f}e9 ({st['aabb_tests']/d['node_trips']:.1f} lanes)  leaf {d['leaf_stages']/1e6:.0f}e6  "
          f"shade {d['shade_stages']/1e6:.1f}e6 ({d['lanes_shaded']/d['shade_stages']:.1f} lanes)  refill {d['refill_stages']/1e6:.1f}e6 ({d['lanes_refilled']/d['refill_stages']:.1f} lanes)  "
          f"wave busy {ss[6]/max(1,ss[8])/max(1,ss[7])*100:.1f}%")
