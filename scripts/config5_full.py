"""BASELINE config 5 whole, at its full 2000 spp (2401x1601 px, final scene + earth-textured sphere + Dielectric sphere + mirror
InfinitePlane, 1.9e10 rays): the HIP frame against the oracle's, every PixelStats and every counter.  The oracle needs about four
minutes on the GPU box's 16 CPUs, which is why this is a script and the pytest suite holds the 100-spp frame and a 2000-spp row."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle as orc  # noqa: E402
import ray_tracing_fsharp_amd as rt  # noqa: E402

earth = np.load(os.path.join(ROOT, "tests", "golden", "earthmap_rgb.npz"))["rgb"]
objs, cam, w, h = rt.sample_images.config5_mixed(earth)
t0 = time.time()
scene = rt.Scene.make(objs)
print("tune:", scene.tune(w, h, cam, seed=5 ^ 0x5EED), flush=True)  # the bench's configuration: tuned walk tree
res = scene.render_rows(w, h, cam, seed=5, counters=True)
plain = scene.render_rows(w, h, cam, seed=5)
print(f"HIP: {res.stats['rays']} rays, kernel {plain.stats['kernel_ms']:.1f} ms (counting variant {res.stats['kernel_ms']:.1f} ms)", flush=True)
t0 = time.time()
acc, rgb, st = orc.OracleScene(objs).render_rows(w, h, cam.to_abi(), seed=5, threads=int(sys.argv[1]) if len(sys.argv) > 1 else 16)
print(f"oracle: {st['rays']} rays in {time.time() - t0:.0f} s", flush=True)
same = np.array_equal(res.accum, acc) and np.array_equal(res.rgb, rgb) and np.array_equal(plain.accum, acc)
keys = ("rays", "prim_tests", "reflections", "samples", "pixels", "pixels_early")
print("PixelStats equal:", same, "| counters equal:", all(res.stats[k] == st[k] for k in keys), {k: st[k] for k in keys})
sys.exit(0 if same and all(res.stats[k] == st[k] for k in keys) else 1)
