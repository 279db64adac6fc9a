"""One-off sanity run of BASELINE config 4's single-GPU share or whole frame: 7681x4321, 1000 spp (prints counters and time)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ray_tracing_fsharp_amd as rt
from ray_tracing_fsharp_amd import distributed as rtd
rt.set_passes(int(os.environ.get('RTFS_PASSES', '0')))
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8   # render rank 0's shard of `world`
objs, cam, w, h = rt.sample_images.config3_final(spp=1000, depth=50, pixels=2160)
w, h = 3840, 2160
scene = rt.Scene.make(objs)
rows, cols = 2*h+1, 2*w+1
first, stride, n = rtd.shard_rows(rows, 0, world)
local = torch.zeros((n, cols, 4), dtype=torch.int32, device="cuda:0")
t0 = time.time()
st = rtd.render_shard_device(scene, cam, w, h, 2024, 0, first, stride, n, local, counters=True, want_stats=True)
a = local.cpu().numpy()
print(f"{cols}x{rows} px, shard 0 of {world}: {n} rows, kernel {st['kernel_ms']:.1f} ms, rays {st['rays']}, samples {st['samples']}, "
      f"Mray/s {st['rays']/st['kernel_ms']/1e3:.0f}, counts {np.unique(a[...,0]).tolist()}, max sum {a[...,1:].max()}, wall {time.time()-t0:.1f}s")
