"""Long-running fuzz of the HIP path against the oracle (run by hand on a GPU box: `python tests/fuzz_campaign.py [n_scenes] [first_seed] [large]`;
the pytest suite runs a 40-scene slice of the same idea).  Scenes: 3-60 bounded spheres of every style, overlapping and nested, now and
then exact duplicates (equal t^2: the tie rule), negative radii, unbounded spheres and planes, cameras anywhere, bounce depths from 0.
Every sixth sphere or so carries a parameterised texture (checkered UV ramps, an image).  Each scene is rendered with the surface-area
walk tree, with the reference's own tree and with the tree tuned to the scene's camera (rt_scene_tune: probe render, ray-count build,
thinning), by the counting kernel variant (compiled node loop) and by the plain one (hand-written node loop, the one the bench
times); all must equal the oracle bit for bit.  With `large` the scenes have 900-6000 smaller spheres: they do not fit the LDS, so the
plain variant is the one whose filter tree is split between LDS and global memory (node_loop_glb32)."""
import dataclasses
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle as orc  # noqa: E402
import scenes  # noqa: E402

rt = scenes.rt
P, V, S, PS, H, Px, Tex = scenes.P, scenes.V, scenes.S, scenes.PS, scenes.H, scenes.Px, scenes.Tex


def scene(seed, large=False):
    rng = np.random.default_rng(seed)
    u = lambda a, b: float(rng.uniform(a, b))  # noqa: E731
    col = lambda: Px(*(int(x) for x in rng.integers(0, 256, 3)))  # noqa: E731

    def style():
        k = int(rng.integers(0, 7))
        return [S.LightSource(Tex(col())), S.LightSourceCap(col()), S.PureReflection(u(0, 1), Tex(col())), S.FuzzedReflection(u(0, 1), Tex(col()), u(0, 1)),
                S.LambertReflection(u(0, 1), Tex(col())), S.Dielectric(u(0, 1), Tex(col()), u(0.7, 2.0), u(0, 1)), S.Glass(u(0.5, 1), Tex(col()), u(0.7, 2.0))][k]

    PT = rt.ParameterisedTexture
    img = scenes.checker_image(seed=seed % 7)

    def textured(c, r):  # a LightSource / Lambert / Pure sphere whose texture is a function of the strike point (Texture.fs:50-67)
        base = [PT.Checkered(PT.UvRamp("u", int(rng.integers(0, 256)), "v"), PT.UvRamp(int(rng.integers(0, 256)), "u", "v"), u(2.0, 80.0)),
                PT.Image(img), PT.UvRamp("v", "u", int(rng.integers(0, 256)))][int(rng.integers(0, 3))]
        tex = PT.toTexture((abs(r), c), base)
        return [S.LightSource(tex), S.LambertReflection(u(0, 1), tex), S.PureReflection(u(0, 1), tex)][int(rng.integers(0, 3))]

    objs = []
    n = int(rng.integers(900, 6001)) if large else int(rng.integers(3, 61))
    spread = u(1.0, 6.0)
    rmax = 0.25 if large else 1.5
    for _ in range(n):
        c, r = P(u(-spread, spread), u(-1, 2.5), u(0, 2 * spread)), u(0.05 * rmax, rmax) * (-1.0 if rng.random() < 0.08 else 1.0)
        objs.append(H.Sphere(rt.Sphere.make(textured(c, r) if rng.random() < (40.0 / n if large else 0.15) else style(), c, r)))
        if rng.random() < 0.1:  # an exact duplicate with another material: equal t^2
            objs.append(H.Sphere(rt.Sphere.make(style(), c, r)))
    for _ in range(int(rng.integers(0, 3))):
        objs.append(H.UnboundedSphere(rt.Sphere.make(style(), P(u(-3, 3), u(-3, 3), u(0, 6)), u(0.5, 3.0))))
    for _ in range(int(rng.integers(0, 3))):
        st = [PS.LightSource(Tex(col())), PS.PureReflection(u(0, 1), col()), PS.LambertReflection(u(0, 1), col()), PS.FuzzedReflection(u(0, 1), col(), u(0, 1))][int(rng.integers(0, 4))]
        objs.append(H.InfinitePlane(rt.InfinitePlane.make(st, P(u(-3, 3), u(-3, 3), u(-3, 6)), scenes.unit(*rng.normal(size=3)))))
    if rng.random() < 0.8:
        objs.append(H.UnboundedSphere(rt.Sphere.make(S.LightSource(Tex(col())), P(0.0, 0.0, 0.0), u(30.0, 400.0))))
    order = rng.permutation(len(objs))
    objs = [objs[i] for i in order]
    cam = dataclasses.replace(rt.Camera.makeBasic(int(rng.integers(1, 40)), u(0.5, 3.0), u(0.8, 2.0), P(u(-2, 2), u(-0.5, 2.0), u(-4, 2)),
                                                  scenes.unit(*rng.normal(size=3)), V(0.0, 1.0, 0.05)), BounceDepth=int(rng.integers(0, 30)))
    return objs, cam, int(rng.integers(2, 13)), int(rng.integers(2, 9))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    large = len(sys.argv) > 3 and sys.argv[3] == "large"
    rays = 0
    for i in range(first, first + n):
        objs, cam, w, h = scene(i, large)
        acc, rgb, st = orc.OracleScene(objs).render_rows(w, h, cam.to_abi(), seed=i, threads=8)
        for tree in ("sah", "reference", "tuned"):
            rt.set_walk_tree("reference" if tree == "reference" else "sah")
            try:
                s = rt.Scene.make(objs)
            finally:
                rt.set_walk_tree("sah")
            if tree == "tuned":
                s.tune(w, h, cam, seed=i)
            for passes in ((0, 2) if i % 5 == 0 else (0,)):
                rt.set_passes(passes)
                try:
                    res = s.render_rows(w, h, cam, seed=i, counters=True)
                    plain = s.render_rows(w, h, cam, seed=i)
                finally:
                    rt.set_passes(0)
                bad = (not np.array_equal(res.accum, acc)) or any(res.stats[k] != st[k] for k in ("rays", "prim_tests", "reflections", "samples"))
                bad = bad or not np.array_equal(plain.accum, acc) or plain.stats["samples"] != st["samples"]
                if tree == "reference":
                    bad = bad or res.stats["aabb_tests"] != st["aabb_tests"]
                if bad:
                    print(f"MISMATCH scene {i} tree {tree} passes {passes}: {np.count_nonzero(np.any(res.accum != acc, axis=-1))} pixels differ", flush=True)
                    sys.exit(1)
        rays += st["rays"]
        if (i - first) % (20 if large else 100) == (19 if large else 99):
            print(f"{i - first + 1} scenes ok, {rays} rays", flush=True)
    print(f"fuzz campaign: {n} {'large ' if large else ''}scenes from seed {first}, {rays} rays, three walk trees (surface-area, reference, tuned): all equal to the oracle", flush=True)


if __name__ == "__main__":
    main()
