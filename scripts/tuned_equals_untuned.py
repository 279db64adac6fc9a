"""BASELINE configs 4 and 5 whole, rendered over the tuned walk tree and over the surface-area tree as built: every PixelStats and every
counter but the box tests must be equal (the untuned frames are the ones scripts/config4_full.py / config5_full.py hold to the oracle)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ray_tracing_fsharp_amd as rt  # noqa: E402

ok = True
for name in ("c4", "c5"):
    if name == "c4":
        objs, cam, _, _ = rt.sample_images.config3_final(seed=2024, spp=1000, depth=50)
        w, h = 3840, 2160
    else:
        earth = np.load(os.path.join(ROOT, "tests", "golden", "earthmap_rgb.npz"))["rgb"]
        objs, cam, w, h = rt.sample_images.config5_mixed(earth)
    a = rt.Scene.make(objs).render_rows(w, h, cam, seed=2024, counters=True)
    s = rt.Scene.make(objs)
    info = s.tune(w, h, cam, seed=2024 ^ 0x5EED)
    b = s.render_rows(w, h, cam, seed=2024, counters=True)
    c = s.render_rows(w, h, cam, seed=2024)
    same = np.array_equal(a.accum, b.accum) and np.array_equal(a.accum, c.accum) and all(a.stats[k] == b.stats[k] for k in ("rays", "prim_tests", "reflections", "samples", "pixels_early"))
    ok = ok and same
    print(f"{name}: {a.stats['rays']} rays, box tests per ray {a.stats['aabb_tests'] / a.stats['rays']:.2f} -> {b.stats['aabb_tests'] / b.stats['rays']:.2f}, "
          f"timed kernel {c.stats['kernel_ms']:.1f} ms, tuned == untuned: {same}", flush=True)
sys.exit(0 if ok else 1)
