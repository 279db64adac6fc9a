"""BASELINE config 4 whole (the final scene at maxW=3840, maxH=2160 -> 7681x4321 px, 1000 spp, 6.0e10 rays): the HIP frame
against the oracle's, every PixelStats and every counter.  The oracle needs about 13 minutes on the GPU box's 16 CPUs; the pytest
suite holds the size-independent properties, three rows against the oracle and a 1-of-8 shard against the frame."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle as orc  # noqa: E402
import ray_tracing_fsharp_amd as rt  # noqa: E402

objs, cam, _, _ = rt.sample_images.config3_final(seed=2024, spp=1000, depth=50)
w, h = 3840, 2160
scene = rt.Scene.make(objs)
print("tune:", scene.tune(w, h, cam, seed=2024 ^ 0x5EED), flush=True)  # the bench's configuration: tuned walk tree, timed kernel variant
res = scene.render_rows(w, h, cam, seed=2024)
print(f"HIP: {res.stats['samples']} samples, kernel {res.stats['kernel_ms']:.1f} ms", flush=True)
t0 = time.time()
rows = 2 * h + 1
acc = np.zeros_like(res.accum)
tot = {}
step = 240  # in slabs, so that progress shows (gpurun takes seven silent minutes for a hang)
for first in range(0, rows, step):
    n = min(step, rows - first)
    a, _, st = orc.OracleScene(objs).render_rows(w, h, cam.to_abi(), seed=2024, row_first=first, row_stride=1, n_rows=n, threads=int(sys.argv[1]) if len(sys.argv) > 1 else 16)
    acc[first:first + n] = a
    for k in ("rays", "samples", "pixels_early"):
        tot[k] = tot.get(k, 0) + st[k]
    print(f"oracle rows {first}..{first + n - 1} done, {time.time() - t0:.0f} s, equal so far: {np.array_equal(res.accum[:first + n], acc[:first + n])}", flush=True)
same = np.array_equal(res.accum, acc)
print("PixelStats equal:", same, "| samples equal:", tot["samples"] == res.stats["samples"], tot)
sys.exit(0 if same and tot["samples"] == res.stats["samples"] else 1)
