"""Kernel time of one workload under the fused kernel, two-pass, and the automatic choice (and a few unit sizes)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ray_tracing_fsharp_amd as rt
from ray_tracing_fsharp_amd import distributed as rtd
which = sys.argv[1] if len(sys.argv) > 1 else "c2"
si = rt.sample_images
objs, cam, w, h = {"c2": lambda: si.config2_three_lambert(), "c3": lambda: si.config3_final()}[which]()
scene = rt.Scene.make(objs)
rows, cols = 2 * h + 1, 2 * w + 1
local = torch.zeros((rows, cols, 4), dtype=torch.int32, device="cuda:0")
ref = None
for passes in (0, 1, 2):
    for chunk in (0, 8, 16, 32):
        rt.set_passes(passes); rt.set_launch_config(0, chunk)
        ts = []
        for _ in range(6):
            st = rtd.render_shard_device(scene, cam, w, h, 2024, 0, 0, 1, rows, local, want_stats=True)
            ts.append(st["total_ms"] if "total_ms" in st else st["kernel_ms"])
        if ref is None: ref = local.clone()
        assert torch.equal(ref, local)
        print(f"passes {passes} chunk {chunk:2d}: {statistics.median(ts[1:]):.3f} ms (kernel_ms {st['kernel_ms']:.3f})", flush=True)
