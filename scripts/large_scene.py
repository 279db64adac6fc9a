"""A scene far beyond the LDS (tests/scenes.many_spheres with 10^5 .. 10^6 spheres): host build time, rt_scene_tune, one frame of the
timed kernel, and the whole image against the oracle.
usage: python scripts/large_scene.py [--n 200000] [--pixels 90] [--spp 16] [--depth 8] [--no-oracle]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import ray_tracing_fsharp_amd as rt  # noqa: E402
import scenes  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=200000)
ap.add_argument("--pixels", type=int, default=90)
ap.add_argument("--spp", type=int, default=16)
ap.add_argument("--depth", type=int, default=8)
ap.add_argument("--no-oracle", action="store_true")
a = ap.parse_args()
objs, cam, w, h = scenes.many_spheres(n=a.n, seed=3, spp=a.spp, depth=a.depth, pixels=a.pixels)
t0 = time.time()
s = rt.Scene.make(objs)
t1 = time.time()
info = s.info()
print({"spheres": a.n, "scene_make_s": round(t1 - t0, 2), "n_nodes": info["n_nodes"], "scene_bytes": info["scene_bytes"], "lds_resident": info["lds_resident"]}, flush=True)
r0 = s.render_rows(w, h, cam, seed=5)  # warm-up (module load, first launch)
t2 = time.time()
r0 = s.render_rows(w, h, cam, seed=5)
t3 = time.time()
tune = s.tune(w, h, cam, seed=5)
t4 = time.time()
r1 = s.render_rows(w, h, cam, seed=5)
t5 = time.time()
cnt = s.render_rows(w, h, cam, seed=5, counters=True)
print({"image": [2 * w + 1, 2 * h + 1], "frame_as_built_s": round(t3 - t2, 3), "tune_s": round(t4 - t3, 3), "frame_tuned_s": round(t5 - t4, 3), "rays": cnt.stats["rays"],
       "box_tests_per_ray": round(cnt.stats["aabb_tests"] / cnt.stats["rays"], 1), "leaf_tests_per_ray": round(cnt.stats["prim_tests"] / cnt.stats["rays"], 2),
       "walk_tree_nodes": s.info()["walk_tree_nodes"]}, flush=True)
assert np.array_equal(r0.accum, r1.accum) and np.array_equal(r0.accum, cnt.accum), "kernel variants or trees disagree"
if not a.no_oracle:
    import oracle as orc  # noqa: E402
    t6 = time.time()
    o = orc.OracleScene(objs)
    t7 = time.time()
    acc, rgb, st = o.render_rows(w, h, cam.to_abi(), seed=5, threads=16)
    t8 = time.time()
    ok = bool(np.array_equal(r1.accum, acc) and np.array_equal(r1.rgb, rgb) and all(cnt.stats[k] == st[k] for k in ("rays", "prim_tests", "reflections", "samples")))
    print({"oracle_make_s": round(t7 - t6, 2), "oracle_frame_s": round(t8 - t7, 2), "oracle_box_tests_per_ray": round(st["aabb_tests"] / st["rays"], 1), "equal": ok}, flush=True)
    assert ok
