"""rt_scene_tune A/B on a BASELINE config: the frame before and after tuning (must be identical), box tests per ray, kernel time."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ray_tracing_fsharp_amd as rt  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "c3"
if which == "c3":
    objs, cam, w, h = rt.sample_images.config3_final()
elif which == "c5":
    earth = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "earthmap_rgb.npz"))["rgb"]
    objs, cam, w, h = rt.sample_images.config5_mixed(earth, spp=200)
else:
    raise SystemExit("c3 | c5")
a = rt.Scene.make(objs)
b = rt.Scene.make(objs)
info = b.tune(w, h, cam, seed=2024)
print("tune:", info, flush=True)
print("info after:", {k: b.info()[k] for k in ("n_nodes", "walk_tree_nodes", "walk_tree", "walk_tree_depth", "scene_bytes", "lds_resident")})
ra = a.render_rows(w, h, cam, seed=2024, counters=True)
rb = b.render_rows(w, h, cam, seed=2024, counters=True)
print("frames equal:", np.array_equal(ra.accum, rb.accum), "| rays", ra.stats["rays"], rb.stats["rays"])
print(f"box tests per ray: {ra.stats['aabb_tests'] / ra.stats['rays']:.2f} -> {rb.stats['aabb_tests'] / rb.stats['rays']:.2f}; prim tests {ra.stats['prim_tests']} {rb.stats['prim_tests']}")
for name, s in (("sah", a), ("tuned", b), ("sah", a), ("tuned", b)):
    t = [s.render_rows(w, h, cam, seed=2024).stats["kernel_ms"] for _ in range(3)]
    print(f"{name}: kernel ms {min(t):.2f} (of {', '.join(f'{x:.2f}' for x in t)})", flush=True)
c = rt.Scene.make(objs)
info2 = c.tune(w, h, cam, seed=2024)
sa, pa, ba = b.walk_tree(); sb, pb, bb = c.walk_tree()
print("same tree from a second tune:", np.array_equal(sa, sb) and np.array_equal(pa, pb) and np.array_equal(ba, bb))
