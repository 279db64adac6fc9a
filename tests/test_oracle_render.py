"""Oracle-level render checks on CPU: the committed golden fixtures still reproduce, the render is independent of thread
count and sharding, adaptive sampling takes the counts Scene.renderPixel prescribes, and analytic scenes give known bytes."""
import dataclasses

import numpy as np
import pytest

import scenes

FIXTURES = {
    "oracle_all_materials_seed0": lambda rt: scenes.all_materials(),
    "oracle_final_thumb_seed7": lambda rt: scenes.small_final(),
    "oracle_config2_small_seed2": lambda rt: rt.sample_images.config2_three_lambert(spp=30, pixels=27),
    "oracle_earth_thumb_seed3": lambda rt: scenes.earth_thumb(scenes.golden("earthmap_rgb")["rgb"]),
}
KEYS = ("rays", "aabb_tests", "prim_tests", "reflections", "samples", "pixels_early")


@pytest.mark.parametrize("name", sorted(FIXTURES))
def test_oracle_reproduces_the_committed_fixtures(rt, orc, name):
    g = scenes.golden(name)
    objs, cam, w, h = FIXTURES[name](rt)
    assert (w, h) == (int(g["max_w"]), int(g["max_h"]))
    acc, rgb, st = orc.OracleScene(objs).render_rows(w, h, cam.to_abi(), seed=int(g["seed"]), threads=4)
    assert np.array_equal(acc, g["accum"]) and np.array_equal(rgb, g["rgb"])
    assert [st[k] for k in KEYS] == g["stats"].tolist()


def test_threads_and_shards_do_not_change_the_image(rt, orc):
    objs, cam, w, h = scenes.all_materials(pixels=12)
    o = orc.OracleScene(objs)
    full, _, st1 = o.render_rows(w, h, cam.to_abi(), seed=11, threads=1)
    full8, _, st8 = o.render_rows(w, h, cam.to_abi(), seed=11, threads=8)
    assert np.array_equal(full, full8) and [st1[k] for k in KEYS] == [st8[k] for k in KEYS]
    for world in (2, 5):
        out = np.zeros_like(full)
        tot = 0
        for rank in range(world):
            part, _, st = o.render_rows(w, h, cam.to_abi(), seed=11, row_first=rank, row_stride=world, threads=2)
            out[rank::world] = part
            tot += st["rays"]
        assert np.array_equal(out, full) and tot == st1["rays"]


@pytest.mark.parametrize("spp", [1, 2, 3, 10, 11, 12, 25])
def test_adaptive_sampling_counts(rt, orc, spp):
    """Scene.renderPixel (Scene.fs:172-194): k = min 5 (spp/2); 2k+1 samples, then spp-2k-1 more unless the integer mean is
    unchanged.  spp = 2 therefore takes THREE samples."""
    objs, cam, w, h = scenes.all_materials(spp=spp, pixels=6)
    acc, rgb, st = orc.OracleScene(objs).render_rows(w, h, cam.to_abi(), seed=1, threads=4)
    k = min(5, spp // 2)
    early, full = 2 * k + 1, max(2 * k + 1, spp)
    assert set(np.unique(acc[..., 0])) <= {early, full}
    assert st["samples"] == int(acc[..., 0].sum())
    assert np.array_equal(rgb, (acc[..., 1:] // acc[..., :1]).astype(np.uint8))  # PixelStats.mean is integer division


def test_empty_scene_and_hot_pink(rt, orc):
    """F2/F3 of SURVEY.md: no sky (a miss is Black); a path that outlives the bounce limit is HotPink (Scene.fs:98,114)."""
    objs, cam, w, h = rt.sample_images.config1_empty()
    acc, rgb, st = orc.OracleScene(objs).render_rows(w, h, cam.to_abi(), seed=1)
    assert acc.shape == (101, 201, 4) and np.all(acc[..., 0] == 1) and not acc[..., 1:].any() and st["rays"] == st["samples"]
    P, H, PS = rt.Point.make, rt.Hittable, rt.InfinitePlaneStyle
    up = rt.Vector.unitise(rt.Vector.make(0.0, 0.0, 1.0))
    mirrors = [H.InfinitePlane(rt.InfinitePlane.make(PS.PureReflection(1.0, rt.Colour.White), P(0.0, 0.0, 5.0), up)),
               H.InfinitePlane(rt.InfinitePlane.make(PS.PureReflection(1.0, rt.Colour.White), P(0.0, 0.0, -5.0), up))]
    cam = dataclasses.replace(rt.Camera.makeBasic(3, 1.0, 1.0, P(0.0, 0.0, 0.0), up, rt.Vector.make(0.0, 1.0, 0.0)), BounceDepth=7)
    acc, rgb, st = orc.OracleScene(mirrors).render_rows(4, 4, cam.to_abi(), seed=3)
    assert np.all(rgb == [205, 105, 180]) and st["rays"] == st["samples"] * 8  # depth+1 hits per path


def test_analytic_mirror_chain_bytes(rt, orc):
    """A camera inside a uniform LightSource dome, looking at a pure mirror plane: every path is mirror -> dome, so every
    sample is exactly combine(darken 0.5 White, dome) = combine((128,128,128), (200,100,50)) (SURVEY.md 8c pin iii)."""
    P, H = rt.Point.make, rt.Hittable
    up = rt.Vector.unitise(rt.Vector.make(0.0, 0.0, 1.0))
    objs = [H.InfinitePlane(rt.InfinitePlane.make(rt.InfinitePlaneStyle.PureReflection(0.5, rt.Colour.White), P(0.0, 0.0, 3.0), up)),
            H.UnboundedSphere(rt.Sphere.make(rt.SphereStyle.LightSource(rt.Texture.Colour(rt.Pixel(200, 100, 50))), P(0.0, 0.0, 0.0), 50.0))]
    cam = rt.Camera.makeBasic(20, 1.0, 1.0, P(0.0, 0.0, 0.0), up, rt.Vector.make(0.0, 1.0, 0.0))
    acc, rgb, st = orc.OracleScene(objs).render_rows(3, 3, cam.to_abi(), seed=9)
    want = [(128 * 200) // 255, (128 * 100) // 255, (128 * 50) // 255]  # darken 0.5 of 255 = 128 (127.5 -> even)
    assert np.all(rgb == want) and np.all(acc[..., 0] == 11) and st["rays"] == 2 * st["samples"]


def test_darken_is_round_half_to_even(orc):
    """Pixel.darken (Pixel.fs:144-151) uses Math.Round = banker's rounding: SURVEY.md Appendix C."""
    got = orc.pixel_darken([[255, 125, 255]], [0.5])[0].tolist()
    assert got == [128, 62, 128]
    assert orc.pixel_darken([[255, 255, 255]], [0.95])[0].tolist() == [242, 242, 242]
    assert orc.pixel_combine([[200, 0, 255]], [[128, 9, 255]])[0].tolist() == [100, 0, 255]


def test_monte_carlo_agrees_with_an_independent_estimate(rt, orc):
    """SURVEY.md 8c pin (iv), statistical: a Lambert floor under a uniform dome.  One bounce off an albedo-a surface lit by a
    dome of byte value L returns about a*L in the limit of many samples (the byte arithmetic re-quantises per bounce, hence
    the tolerance); multiple floor bounces cannot happen (the floor is convex to the dome)."""
    P, H, S = rt.Point.make, rt.Hittable, rt.SphereStyle
    objs = [H.UnboundedSphere(rt.Sphere.make(S.LambertReflection(0.5, rt.Texture.Colour(rt.Colour.White)), P(0.0, -1000.0, 0.0), 1000.0)),
            H.UnboundedSphere(rt.Sphere.make(S.LightSource(rt.Texture.Colour(rt.Pixel(200, 200, 200))), P(0.0, 0.0, 0.0), 5000.0))]
    down = rt.Vector.unitise(rt.Vector.make(0.0, -1.0, 0.2))
    cam = rt.Camera.makeBasic(200, 1.0, 1.0, P(0.0, 2.0, 0.0), down, rt.Vector.make(0.0, 0.0, 1.0))
    acc, rgb, st = orc.OracleScene(objs).render_rows(2, 2, cam.to_abi(), seed=4)
    mean = acc[..., 1] / acc[..., 0]
    assert np.all(np.abs(mean - 100.0) <= 1.0)  # darken 0.5 of 255 = 128; combine(128, 200) = 100


def test_tree_walk_agrees_with_brute_force(rt, orc):
    """SURVEY.md 8c pin (iv): the unpruned tree walk finds the same closest hit as testing every sphere.  They may differ only
    for rays grazing a sphere inside Sphere.firstIntersection's 1e-8 discriminant band but outside its box, so: identical
    closest objects on 100k random rays up to a handful, and near-identical renders."""
    objs, cam, w, h = scenes.small_final(spp=24, pixels=12)
    o = orc.OracleScene(objs)
    rays = scenes.random_rays(100000, 5, origin_scale=5.0)
    rays[:50000, :3] = [13.0, 2.0, -3.0]
    hit_tree, strike_tree, cnt_tree = o.hit_object(rays)
    acc_tree, _, st_tree = o.render_rows(w, h, cam.to_abi(), seed=3, threads=4)
    try:
        orc.set_brute_force(True)
        hit_bf, strike_bf, cnt_bf = o.hit_object(rays)
        acc_bf, _, st_bf = o.render_rows(w, h, cam.to_abi(), seed=3, threads=4)
    finally:
        orc.set_brute_force(False)
    assert np.count_nonzero(hit_tree != hit_bf) <= 3
    same = hit_tree == hit_bf
    assert np.array_equal(strike_tree[same & (hit_tree >= 0)], strike_bf[same & (hit_bf >= 0)])
    assert cnt_bf[:, 0].sum() == 0 and cnt_bf[:, 1].min() >= 485  # no boxes; every sphere tested
    assert np.mean(np.any(acc_tree != acc_bf, axis=-1)) < 0.01
    assert abs(int(st_tree["rays"]) - int(st_bf["rays"])) < 0.01 * st_tree["rays"]


def test_order_of_bounded_objects_does_not_change_a_pixel(orc):
    """BoundingBoxTree.make's shape depends on the input order, the closest hit does not (barring exact t^2 ties, which distinct
    random spheres do not produce): permuting the Hittable array leaves every PixelStats unchanged.  (The unbounded objects keep
    their relative order: Scene.hitObject compares them with a tolerance, first come first served.)"""
    import numpy as np
    import scenes
    objs, cam, w, h = scenes.small_final(seed=11, spp=16, depth=20, pixels=10)
    base, _rgb, st0 = orc.OracleScene(objs).render_rows(w, h, cam.to_abi(), seed=4, threads=4)
    rng = np.random.default_rng(8)
    bounded = [i for i, o in enumerate(objs) if o.kind == 0]
    for _ in range(3):
        perm = list(range(len(objs)))
        shuffled = rng.permutation(bounded)
        for dst, src in zip(bounded, shuffled):
            perm[dst] = int(src)
        acc, _rgb, st = orc.OracleScene([objs[i] for i in perm]).render_rows(w, h, cam.to_abi(), seed=4, threads=4)
        assert np.array_equal(acc, base)
        assert (st["rays"], st["prim_tests"], st["reflections"]) == (st0["rays"], st0["prim_tests"], st0["reflections"])
