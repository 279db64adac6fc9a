import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import ray_tracing_fsharp_amd as rt
from ray_tracing_fsharp_amd import distributed as rtd
objs, cam, w, h = rt.sample_images.config3_final(spp=500, depth=50, pixels=int(sys.argv[1]) if len(sys.argv) > 1 else 200)
scene = rt.Scene.make(objs)
rows, cols = 2*h+1, 2*w+1
def render(counters=False):
    local = torch.zeros((rows, cols, 4), dtype=torch.int32, device="cuda:0")
    st = rtd.render_shard_device(scene, cam, w, h, 2024, 0, 0, 1, rows, local, counters=counters, want_stats=True)
    return local.cpu().numpy(), st
for passes in (0, 1, 2):
    for park in (0, -1):
        rt.set_passes(passes); rt.set_park(park)
        ref, st0 = render(True)
        diffs = []
        for i in range(4):
            a, st = render(False)
            diffs.append((int(np.count_nonzero(np.any(a != ref, axis=-1))), st["samples"] - st0["samples"]))
        print("passes", passes, "park", park, "differing pixels / sample delta per launch:", diffs, flush=True)
