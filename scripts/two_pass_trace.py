"""A few two-pass renders of a small workload, for `rocprofv3 --kernel-trace`: where does the time between pass A and pass B go?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ray_tracing_fsharp_amd as rt
from ray_tracing_fsharp_amd import distributed as rtd
objs, cam, w, h = rt.sample_images.config2_three_lambert()
scene = rt.Scene.make(objs)
rows, cols = 2 * h + 1, 2 * w + 1
local = torch.zeros((rows, cols, 4), dtype=torch.int32, device="cuda:0")
rt.set_passes(int(sys.argv[1]) if len(sys.argv) > 1 else 2)
for _ in range(5):
    st = rtd.render_shard_device(scene, cam, w, h, 2024, 0, 0, 1, rows, local, want_stats=True)
    print(st["kernel_ms"], st.get("total_ms"), flush=True)
