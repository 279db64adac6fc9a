"""Host logic of the product (scene flattening, tree build, camera, PPM, scene catalogue) against the oracle's
independent restatement and the known answers of SURVEY.md Appendix C.  CPU only."""
import dataclasses
import os

import numpy as np
import pytest

import scenes

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("which", ["final", "all_materials", "thumb", "one", "two", "three", "threaded"])
def test_flattened_tree_equals_the_pointer_tree(rt, orc, which):
    """BoundingBoxTree.make (BoundingBoxTree.fs:9-43): the product's index-array build against the oracle's recursive
    pointer build -- same pre-order, same skip links, same leaf hittables, bit-identical boxes."""
    if which == "final":
        objs = rt.sample_images.config3_final()[0]
    elif which == "all_materials":
        objs = scenes.all_materials()[0]
    elif which == "thumb":
        objs = scenes.small_final(seed=99)[0]
    elif which == "threaded":  # from 4096 leaves the product builds axes and subtrees on threads (rt_scene.h): the same tree
        objs = scenes.many_spheres(n=40000, seed=8)[0]
        objs[100:140] = [objs[100]] * 40  # equal sort keys on every axis: the stable order decides
    else:
        n = {"one": 1, "two": 2, "three": 3}[which]
        objs = scenes.small_final()[0][:n]
    s, o = rt.Scene.make(objs), orc.OracleScene(objs)
    sk, pr, bx = s.tree()
    sk2, pr2, bx2, depth = o.tree()
    assert np.array_equal(sk, sk2) and np.array_equal(pr, pr2)
    assert np.array_equal(bx.view(np.uint64), bx2.view(np.uint64))
    info = s.info()
    nb = sum(1 for h in objs if h.kind == rt._abi.RT_HITTABLE_SPHERE)
    assert info["n_bounded"] == nb and info["n_nodes"] == max(0, 2 * nb - 1) and info["tree_depth"] == depth
    assert info["n_unbounded"] == len(objs) - nb
    if nb:
        leaves = pr[pr >= 0]
        assert sorted(leaves.tolist()) == [i for i, h in enumerate(objs) if h.kind == rt._abi.RT_HITTABLE_SPHERE]
        assert sk[0] == len(sk) and np.all(sk > np.arange(len(sk)))


def test_negative_radius_gives_an_inverted_box(rt):
    """Sphere.make (Sphere.fs:333-336) with radius < 0: Min = c + |r|, Max = c - |r| (kept, not "fixed")."""
    objs = [rt.Hittable.Sphere(rt.Sphere.make(rt.SphereStyle.Glass(1.0, rt.Texture.Colour(rt.Colour.White), 1.5), rt.Point.make(1.0, 2.0, 3.0), -0.5))]
    _, _, bx = rt.Scene.make(objs).tree()
    assert bx[0].tolist() == [1.5, 0.5, 2.5, 1.5, 3.5, 2.5]


def test_final_scene_camera_known_answers(rt, orc):
    """Camera.makeBasic (Camera.fs:34-59) for SampleImages.randomSpheres."""
    _, cam, w, h = rt.sample_images.config3_final()
    c = cam.to_abi()
    assert (w, h) == (1200, 800) and cam.SamplesPerPixel == 500 and cam.BounceDepth == 50
    assert list(c.view_dir) == [-0.9636241116594315, -0.14824986333222023, 0.22237479499833035]
    assert list(c.xaxis_origin) == [3.3637588834056853, 0.5175013666777977, -0.7762520500166965]
    assert list(c.xaxis_dir) == [0.22485950669875845, 0.0, 0.97439119569462]
    assert list(c.yaxis_dir) == [-0.14445336159384609, 0.9889499370655616, 0.033335391137041356]
    assert (c.viewport_width, c.viewport_height) == (3.0, 2.0)
    assert rt.Camera.makeBasic(7, 1.0, 2.0, rt.Point.make(0, 0, 0), rt.Vector.make(0.0, 0.0, 1.0), rt.Vector.make(0.0, 1.0, 0.0)).BounceDepth == 150


def test_camera_matches_the_oracle_bit_for_bit(rt, orc):
    rng = np.random.default_rng(1)
    for _ in range(300):
        o, d, up = rng.normal(size=3) * 5, rng.normal(size=3), rng.normal(size=3)
        d = rt.Vector.unitise(rt.Vector.make(*d))
        if abs(np.dot(d, up / np.linalg.norm(up))) > 0.99:
            continue
        a = rt.Camera.makeBasic(10, abs(rng.normal()) + 0.1, 1.5, rt.Point.make(*o), d, rt.Vector.make(*up)).to_abi()
        b = orc.camera_make_basic(10, a.focal_length, 1.5, o, d, up)
        assert bytes(a) == bytes(b)


def test_aspect_times_pixels_truncates(rt):
    """`aspectRatio * (float pixels) |> int` (SampleImages.fs:96 ...): SURVEY.md section 7's list."""
    si = rt.sample_images
    assert si._extent(16.0 / 9.0, 200) == (355, 200)
    assert si._extent(16.0 / 9.0, 300) == (533, 300)
    assert si._extent(16.0 / 9.0, 400) == (711, 400)
    assert si._extent(16.0 / 9.0, 225) == (400, 225)
    assert si._extent(3.0 / 2.0, 800) == (1200, 800)


def test_random_spheres_recipe(rt):
    """SampleImages.randomSpheres (SampleImages.fs:812-960): counts, mix and fixed objects; deterministic in `seed`."""
    A = rt._abi
    objs, cam, w, h = rt.sample_images.randomSpheres(seed=2024)
    objs2 = rt.sample_images.randomSpheres(seed=2024)[0]
    assert objs == objs2 and objs != rt.sample_images.randomSpheres(seed=2025)[0]
    small = [o for o in objs if o.sphere.Radius == 0.2]
    assert 470 <= len(small) <= 484 and len(objs) == len(small) + 5
    kinds = [o.sphere.Style.style for o in small]
    frac = lambda k: kinds.count(k) / len(small)  # noqa: E731
    assert 0.72 < frac(A.RT_SPHERE_LAMBERT_REFLECTION) < 0.88 and 0.09 < frac(A.RT_SPHERE_FUZZED_REFLECTION) < 0.21
    assert 0.01 < frac(A.RT_SPHERE_GLASS) < 0.10
    for o in small:
        c = o.sphere.Centre
        assert c.y == 0.2 and (c.x - 4.0) ** 2 + (c.z - 0.0) ** 2 > 0.81
    assert [o.kind for o in objs[-2:]] == [A.RT_HITTABLE_UNBOUNDED_SPHERE] * 2
    assert objs[-2].sphere.Radius == 2000.0 and objs[-1].sphere.Centre == (0.0, -1000.0, 0.0)
    assert cam.BounceDepth == 150 and (w, h) == (1200, 800)


def test_catalogue_builds_and_flattens(rt):
    for name in rt.sample_images.CATALOGUE:
        objs, cam, w, h = rt.sample_images.get(name)()
        info = rt.Scene.make(objs).info()
        assert info["n_bounded"] + info["n_unbounded"] == len(objs) and w > 0 and h > 0
    with pytest.raises(ValueError, match="Unrecognised arg"):
        rt.sample_images.get("nope")
    g = rt.sample_images.gradient()
    assert g.shape == (256, 256, 3) and g[3, 5].tolist() == [5, 252, 63]


def test_python_float_producer_is_the_reference_generator(rt, orc):
    st = (123456789, 362436069, 521288629, 88675123)
    p = rt.FloatProducer(st)
    assert [p.Get() for _ in range(500)] == orc.float_producer(st, 500).tolist()


def test_ppm_writer_golden_file_and_gamma(rt, orc, tmp_path):
    """ImageOutput.writePpm (ImageOutput.fs:163-197) against the reference's golden file; PixelOutput.correct
    (ImageOutput.fs:11-18) against SURVEY.md Appendix C and the oracle for all 256 bytes."""
    expected = open(os.path.join(HERE, "golden", "PpmOutputExample.txt"), "rb").read().replace(b"\r\n", b"\n")
    image = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255]], [[255, 255, 0], [255, 255, 255], [0, 0, 0]]], np.uint8)
    assert rt.ImageOutput.formatPpm(False, image) == expected
    out = tmp_path / "o.ppm"
    ticks = []
    rt.ImageOutput.writePpm(False, ticks.append, image, str(out))
    assert out.read_bytes() == expected and len(ticks) == 6
    assert not expected.endswith(b"\n")
    for b, want in ((0, 0), (1, 16), (63, 127), (64, 128), (128, 181), (200, 226), (254, 254), (255, 255)):
        assert rt.PixelOutput.correct(b) == want
    assert [rt.PixelOutput.correct(b) for b in range(256)] == [orc.gamma_correct(b) for b in range(256)]
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, size=(7, 5, 3), dtype=np.uint8)
    for gamma in (False, True):
        assert rt.ImageOutput.formatPpm(gamma, img) == orc.format_ppm(img, gamma=gamma)
    ramp = rt.sample_images.gradient()
    txt = rt.ImageOutput.formatPpm(False, ramp).split(b"\n")
    assert txt[:3] == [b"P3", b"256 256", b"255"] and txt[3].startswith(b"0 255 63 1 255 63")


def test_scene_render_is_lazy_like_the_reference(rt):
    """Scene.render (Scene.fs:196-236) returns (rows as progress, Image) without doing any work."""
    objs, cam, w, h = rt.sample_images.config1_empty()
    total, image = rt.Scene.render(lambda _: None, lambda _: None, w, h, cam, rt.Scene.make(objs))
    assert total == 101.0 and rt.Image.rowCount(image) == 101 and rt.Image.colCount(image) == 201
    assert image._rows is None
    made = rt.Image.make(2, 3, np.zeros((2, 3, 3), np.uint8))
    assert rt.Image.render(made).shape == (2, 3, 3)


def test_camera_with_syntax(rt):
    _, cam, _, _ = rt.sample_images.randomSpheres()
    c2 = dataclasses.replace(cam, BounceDepth=50, SamplesPerPixel=7)
    abi = c2.to_abi()
    assert (abi.bounce_depth, abi.samples_per_pixel) == (50, 7) and cam.to_abi().bounce_depth == 150


def test_lds_residency_is_reported(rt):
    assert rt.Scene.make(rt.sample_images.config3_final()[0]).info()["lds_resident"] == 1
    big = rt.Scene.make(scenes.many_spheres(n=2600)[0]).info()
    assert big["lds_resident"] == 0 and big["scene_bytes"] > 163840


def test_leaf_box_implied_flag_follows_the_scene_extent(rt):
    """rt_scene_info.leaf_box_implied: set only for scenes inside the bounds under which the timed kernel's leaf pass may take a sphere
    hit as proof of the leaf-box hit (csrc/rt_device.h, leaf_test_object_exact): every bounded radius in (0, 100], coordinates within
    1000, r_max * extent <= 500."""
    S, H, P, Tex, Px = rt.SphereStyle, rt.Hittable, rt.Point.make, rt.Texture.Colour, rt.Pixel
    lam = S.LambertReflection(0.5, Tex(Px(9, 9, 9)))

    def flag(spheres):
        return rt.Scene.make([H.Sphere(rt.Sphere.make(lam, P(*c), r)) for c, r in spheres]).info()["leaf_box_implied"]

    three = [((0.0, 0.0, 0.0), 1.0), ((3.0, 0.0, 0.0), 0.5), ((0.0, 4.0, 0.0), 0.2)]
    assert flag(three) == 1
    assert rt.Scene.make(rt.sample_images.config3_final()[0]).info()["leaf_box_implied"] == 1
    assert flag(three + [((0.0, 0.0, 9.0), -0.5)]) == 0            # a negative radius: inverted box (Sphere.fs:333-336)
    assert flag(three + [((0.0, 0.0, 300.0), 150.0)]) == 0         # radius beyond 100
    assert flag(three + [((1200.0, 0.0, 0.0), 1.0)]) == 0          # coordinates beyond 1000
    assert flag(three + [((40.0, 0.0, 0.0), 20.0)]) == 0           # r_max * extent = 20 * 60 > 500
    assert flag([((0.0, 0.0, 0.0), 20.0), ((1.0, 0.0, 0.0), 1.0), ((0.0, 1.0, 0.0), 1.0)]) == 1  # 20 * 20 = 400


def _check_filter_tree(scene):
    """rt_scene_get_filter_tree against rt_scene_get_walk_tree: same tree, records in depth order, outward-rounded boxes."""
    import filter_cases
    skip, prim, boxes = scene.walk_tree()
    fb, lk = scene.filter_tree()
    n = len(skip)
    assert fb.shape == (n, 6) and lk.shape == (n, 5)
    if n == 0:
        return
    assert np.all((lk[:, :2] >= 0) & (lk[:, :2] <= n))
    # "every box hit" visits the records in pre-order: that walk gives each pre-order index its position in the image
    place = np.full(n + 1, -1, np.int64)
    place[n] = n
    pos, seen = 0, np.zeros(n, bool)
    for i in range(n):
        assert 0 <= pos < n and not seen[pos], "links do not walk the tree in pre-order"
        place[i], seen[pos] = pos, True
        pos = int(lk[pos, 0])
    assert pos == n and place[0] == 0                                   # the root is the first record; the walk ends at "exhausted"
    assert sorted(place[:n].tolist()) == list(range(n))                 # a permutation: every record is reached exactly once
    order = np.argsort(place[:n])                                       # pre-order index of the record at each position
    leaf = prim >= 0
    assert np.array_equal(lk[place[:n], 4], prim)                       # the same leaves, as hittable indices
    assert np.array_equal(lk[place[:n], 1], place[skip])                # a miss skips the subtree
    assert np.array_equal(lk[place[:n], 0][leaf], place[skip][leaf])    # a Leaf's hit link: its test is queued, the walk goes on
    assert np.array_equal(lk[place[:n], 0][~leaf], place[1:][~leaf])    # a Branch's: its first child
    entry = lk[place[:n], 2].astype(np.int64) & 0xFFFFFFFF
    assert np.all(entry[~leaf] == 0) and np.all(entry[leaf] != 0)
    assert np.all(lk[place[:n], 3][leaf] == 16) and np.all(lk[place[:n], 3][~leaf] == 0)
    # depth never decreases along the image: the top of the tree is a prefix
    depth = np.zeros(n, np.int64)
    stack = []
    for i in range(n):
        while stack and i >= stack[-1]:
            stack.pop()
        depth[i] = len(stack)
        if not leaf[i]:
            stack.append(int(skip[i]))
    assert np.all(np.diff(depth[order]) >= 0)
    same = np.diff(depth[order]) == 0
    assert np.all(np.diff(order)[same] > 0)                             # pre-order within a level
    # boxes: the walk tree's, rounded outward once
    assert np.array_equal(fb[place[:n]][:, 0::2], filter_cases.f32_down(boxes[:, 0::2]))
    assert np.array_equal(fb[place[:n]][:, 1::2], filter_cases.f32_up(boxes[:, 1::2]))


def test_filter_tree_is_the_walk_tree_in_depth_order(rt):
    """The timed kernel's records (rt_scene_get_filter_tree): for the bench scene as built and tuned, a scene beyond the LDS, one beyond
    16384 objects (full-width queue entries), and the degenerate sizes."""
    objs, cam, w, h = rt.sample_images.config3_final()
    s = rt.Scene.make(objs)
    _check_filter_tree(s)
    rays = scenes.random_rays(30000, 5, origin_scale=6.0)
    s.tune_rays(rays)
    assert s.info()["walk_tree"] == 2
    _check_filter_tree(s)
    big = rt.Scene.make(scenes.many_spheres(n=2600)[0])
    _check_filter_tree(big)
    big.tune_rays(scenes.random_rays(30000, 6, origin_scale=8.0))
    _check_filter_tree(big)
    wide = rt.Scene.make(scenes.many_spheres(n=17000)[0])
    fb, lk = wide.filter_tree()
    assert np.all((lk[lk[:, 4] >= 0, 2].astype(np.int64) & 0x80000000) != 0)
    _check_filter_tree(wide)
    for n in (0, 1, 2, 3):
        _check_filter_tree(rt.Scene.make(scenes.many_spheres(n=n)[0]))
