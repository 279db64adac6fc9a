"""The reference's OWN unit tests (RayTracing.Test/*.fs), replayed against the CPU oracle: this is what pins the oracle.
Every test names the reference test it restates.  FsCheck generators are replaced by seeded numpy generators over the
same kinds of values (NormalFloat -> normal(0, scale)); known-answer cases are copied literally."""
import math
import os

import numpy as np
import pytest

import scenes

DELTA = 0.00000001


def feq(a, b):
    return abs(a - b) < DELTA  # Float.equal (Float.fs:84)


# ---- TestSphereIntersection.fs -------------------------------------------------------------------------------------
def test_intersection_of_sphere_and_ray_lies_on_both_case_1(orc):
    """TestSphereIntersection.fs:37-58 -- the literal case, plus the hand evaluation of SURVEY.md 8c.1."""
    ray = orc.ray_make((1.462205539, -4.888279676, 7.123293244), (-9.549697616, 4.400018428, 10.41024923))
    centre, radius = (-5.688391601, -5.360125644, 9.074300761), 8.199747973
    t = orc.sphere_first_intersection([ray], [list(centre) + [radius]])[0]
    assert t == 12.649517791881394
    assert orc.sphere_lies_on(orc.ray_walk_along(ray, t), centre, radius)


def test_intersection_of_sphere_and_ray_does_lie_on_both(orc):
    """TestSphereIntersection.fs:21-34 (property)."""
    rng = np.random.default_rng(1)
    rays = scenes.random_rays(20000, 2, origin_scale=2.0)
    sph = np.concatenate([rng.normal(size=(20000, 3)) * 2, rng.normal(size=(20000, 1)) * 2], axis=1)
    t = orc.sphere_first_intersection(rays, sph)
    hit = ~np.isnan(t)
    assert hit.sum() > 2000
    for i in np.nonzero(hit)[0]:
        p = orc.ray_walk_along(tuple(rays[i]), float(t[i]))
        assert orc.sphere_lies_on(p, tuple(sph[i, :3]), float(sph[i, 3])), i
    assert np.all(t[hit] > DELTA)  # "Does not return any intersections which are behind us"


# ---- TestSphere.fs ----------------------------------------------------------------------------------------------------
def test_point_at_distance_r_from_centre_lies_on_sphere(orc):
    """TestSphere.fs:14-50."""
    rng = np.random.default_rng(3)
    for _ in range(2000):
        c = rng.normal(size=3)
        r, th, ph = abs(rng.normal()), rng.normal(), rng.normal()
        p = c + np.array([r * math.cos(ph) * math.sin(th), r * math.sin(ph) * math.sin(th), r * math.cos(th)])
        assert orc.sphere_lies_on(tuple(p), tuple(c), r)


def _one_sphere_reflection(rt, orc, style, centre, radius, states):
    objs = [rt.Hittable.Sphere(rt.Sphere.make(style, rt.Point.make(*centre), radius))]
    o = orc.OracleScene(objs)
    n = len(states)
    ray = np.tile([0.0, 0.0, 0.0, 0.0, 0.0, 1.0], (n, 1))
    return o.reflection(np.zeros(n, np.int32), ray, np.tile([255, 255, 255], (n, 1)), np.tile([0.0, 0.0, 1.0], (n, 1)), states)


@pytest.mark.parametrize("case", ["glass_edge", "glass_middle", "dielectric_middle"])
def test_glass_and_dielectric_known_answers(rt, orc, case):
    """TestSphere.fs:53-84 (Glass perfectly reflects against the edge), :87-118 (Glass refracts through the middle),
    :121-152 (Dielectric prob 1.0 through the middle): not absorbed, colour = Green, origin = strike, direction kept."""
    green = rt.Texture.Colour(rt.Colour.Green)
    style, centre = {
        "glass_edge": (rt.SphereStyle.Glass(1.0, green, 1.5), (0.0, 1.0, 1.0)),
        "glass_middle": (rt.SphereStyle.Glass(1.0, green, 1.5), (0.0, 0.0, 2.0)),
        "dielectric_middle": (rt.SphereStyle.Dielectric(1.0, green, 1.5, 1.0), (0.0, 0.0, 2.0)),
    }[case]
    states = np.random.default_rng(4).integers(1, 2 ** 31 - 1, size=(500, 4), dtype=np.uint32)  # `Random () |> FloatProducer`
    absorbed, col, ray, _ = _one_sphere_reflection(rt, orc, style, centre, 1.0, states)
    assert not absorbed.any()
    assert np.all(col == [0, 255, 0])
    assert np.all(np.abs(ray[:, :3] - [0.0, 0.0, 1.0]) < DELTA)  # Point.equal origin strikePoint
    keeps = np.ones(len(states), bool)
    if case == "glass_middle":
        # Through the centre cos = 1, so Schlick's reflectionProb is R0 = ((1-1.5)/(1+1.5))^2 = 0.04 (Sphere.fs:281-292): the
        # reference's test draws from an unseeded FloatProducer and therefore fails one run in 25.  Here both branches are
        # pinned by the actual draw: u < R0 mirrors straight back, otherwise the direction is kept.
        u = np.array([orc.float_producer(st, 1)[0] for st in states])
        keeps = ~(u < 0.04000000000000001)
        assert 5 < (~keeps).sum() < 45
        assert np.all(np.abs(ray[~keeps, 3:] - [0.0, 0.0, -1.0]) < DELTA)
    assert np.all(np.abs(ray[keeps, 3:] - [0.0, 0.0, 1.0]) < DELTA)  # Vector.equal direction


def test_plane_map_round_trip(orc):
    """TestSphere.fs:155-194: planeMapInverse (planeMap theta phi) = (theta, phi) for theta, phi in [0, 1]."""
    rng = np.random.default_rng(5)
    n = 0
    while n < 3000:
        c, r = rng.normal(size=3), abs(rng.normal())
        th, ph = abs(rng.normal()), abs(rng.normal())
        if th > 1.0 or ph > 1.0 or r < 1e-3:
            continue
        n += 1
        p = orc.plane_map(r, tuple(c), th, ph)
        a, b = orc.plane_map_inverse(r, tuple(c), p)
        # near the poles / the seam the map is not injective; the reference's generator hits those with probability ~0
        if min(ph, 1.0 - ph) < 1e-6 or min(th, 1.0 - th) < 1e-6:
            continue
        assert abs(th - a) < 1e-7 and abs(ph - b) < 1e-7, (c, r, th, ph, a, b)


def test_specific_plane_map_inverses(orc):
    """TestSphere.fs:197-204: exact equality (shouldEqual) on the six axis points."""
    f = lambda p: orc.plane_map_inverse(1.0, (0.0, 0.0, 0.0), p)  # noqa: E731
    assert f((1.0, 0.0, 0.0)) == (0.5, 0.5)
    assert f((-1.0, 0.0, 0.0)) == (0.0, 0.5)
    assert f((0.0, 1.0, 0.0)) == (0.5, 1.0)
    assert f((0.0, -1.0, 0.0)) == (0.5, 0.0)
    assert f((0.0, 0.0, 1.0)) == (0.25, 0.5)
    assert f((0.0, 0.0, -1.0)) == (0.75, 0.5)


def test_specific_plane_maps(orc):
    """TestSphere.fs:207-214: Point.equal within 1e-8."""
    f = lambda a, b: orc.plane_map(1.0, (0.0, 0.0, 0.0), a, b)  # noqa: E731
    for (a, b), want in (((0.5, 0.5), (1, 0, 0)), ((0.0, 0.5), (-1, 0, 0)), ((0.5, 1.0), (0, 1, 0)), ((0.5, 0.0), (0, -1, 0)),
                         ((0.25, 0.5), (0, 0, 1)), ((0.75, 0.5), (0, 0, -1))):
        got = f(a, b)
        assert all(feq(g, float(w)) for g, w in zip(got, want)), (a, b, got)


# ---- TestBoundingBox.fs --------------------------------------------------------------------------------------------
def _sort(x1, x2):  # TestBoundingBox.fs:13-14
    return min(x1, x2), (x1 + (DELTA / 2.0) if x1 == x2 else max(x1, x2))


@pytest.mark.parametrize("axis", [0, 1, 2])
@pytest.mark.parametrize("negate", [True, False])
def test_bounding_box_behind_the_ray_is_not_hit(orc, axis, negate):
    """TestBoundingBox.fs:16-43 (x), :45-72 (y), :86-114 (z): a box strictly on the far side of the origin, ray pointing away."""
    rng = np.random.default_rng(10 + axis * 2 + negate)
    n = 5000
    a, b = rng.normal(size=(n, 3)), rng.normal(size=(n, 3))
    lo, hi = np.zeros((n, 3)), np.zeros((n, 3))
    for i in range(n):
        for k in range(3):
            if k == axis:
                u, v = (abs(a[i, k]), abs(b[i, k])) if negate else (-abs(a[i, k]), -abs(b[i, k]))
            else:
                u, v = a[i, k], b[i, k]
            lo[i, k], hi[i, k] = _sort(u, v)
    origin, d = [0.0, 0.0, 0.0], [0.0, 0.0, 0.0]
    origin[axis] = -DELTA if negate else DELTA
    d[axis] = -1.0 if negate else 1.0
    rays = np.tile(origin + d, (n, 1))
    assert not orc.bbox_hits(rays, np.concatenate([lo, hi], axis=1)).any()


def test_bounding_box_forward_ray_going_backward_case_1(orc):
    """TestBoundingBox.fs:74-84: the degenerate zero-thickness box."""
    z1, z2 = _sort(-abs(0.0), -abs(0.0))
    x1, x2 = _sort(0.0, 0.0)
    y1, y2 = _sort(0.0, 1.0)
    assert orc.bbox_hits([[0.0, 0.0, DELTA, 0.0, 0.0, 1.0]], [[x1, y1, z1, x2, y2, z2]])[0] == 0


def test_bounding_box_forward_does_intersect_ray_going_forward(orc):
    """TestBoundingBox.fs:116-123: unit cube hit from inside (inverse directions are +inf, products NaN)."""
    assert orc.bbox_hits([[0.0, 0.0, 0.0, 0.0, 0.0, 1.0]], [[-1.0, -1.0, -1.0, 1.0, 1.0, 1.0]])[0] == 1


# ---- TestPixel.fs ---------------------------------------------------------------------------------------------------
def test_combine_pixels_with_white_and_black(orc):
    """TestPixel.fs:157-169 (combine White = id) and :172-183 (combine Black = Black), exhaustively over one channel."""
    g = np.arange(256, dtype=np.uint8)
    px = np.stack([g, g[::-1], (g.astype(np.int32) * 7 % 256).astype(np.uint8)], axis=1)
    assert np.array_equal(orc.pixel_combine(px, np.full_like(px, 255)), px)
    assert np.array_equal(orc.pixel_combine(px, np.zeros_like(px)), np.zeros_like(px))


# ---- TestRandom.fs ---------------------------------------------------------------------------------------------------
def _seeds(n, seed):
    return np.random.default_rng(seed).integers(0, 2 ** 31 - 1, size=(n, 4), dtype=np.uint32)  # uint (rand.Next ())


def test_random_floats_are_in_the_right_range(orc):
    """TestRandom.fs:14-36: normal (non-zero, finite), not < 0, not > 1."""
    for st in _seeds(50, 20):
        r = orc.float_producer(st, 300)
        assert np.all(np.isfinite(r)) and np.all(r >= 0.0) and np.all(r <= 1.0)
        assert np.all(r > 2.2250738585072014e-308)  # Double.IsNormal


def test_random_floats_are_distributed_over_the_whole_range(orc):
    """TestRandom.fs:39-50: 100 draws cover all ten deciles."""
    for st in _seeds(50, 21):
        r = orc.float_producer(st, 100)
        for i in range(10):
            assert np.any((i * 0.1 < r) & (r < (i + 1) * 0.1)), (st, i)


def test_floats_are_not_obviously_correlated(orc):
    """TestRandom.fs:55-71: consecutive draws are distinct."""
    for st in _seeds(50, 22):
        r = orc.float_producer(st, 6)
        assert len(set(r.tolist())) == 6


def test_float_producer_known_answer(orc):
    """SURVEY.md Appendix C (hand-derived from Float.fs:14-29): state (1,2,3,4)."""
    r = orc.float_producer((1, 2, 3, 4), 4)
    assert r.tolist() == [0.05090332032435185, 0.1214599609657796, 0.01562500000363798, 0.12548828127921752]


# ---- TestRay.fs -------------------------------------------------------------------------------------------------------
def test_walk_along_properties(orc):
    """TestRay.fs:12-82: parallel rays keep their offset, walkAlong walks the right distance, and stays on the ray."""
    rng = np.random.default_rng(30)
    for _ in range(3000):
        o1, o2 = rng.normal(size=3), rng.normal(size=3)
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        m = rng.normal()
        w1 = np.array(orc.ray_walk_along(tuple(o1) + tuple(d), m))
        w2 = np.array(orc.ray_walk_along(tuple(o2) + tuple(d), m))
        assert np.all(np.abs((w1 - w2) - (o1 - o2)) < DELTA)
        assert feq(float(np.dot(w1 - o1, w1 - o1)), m * m)
        if np.min(np.abs(d)) > 1e-3:  # liesOn divides by each component (Ray.fs:56-66)
            assert orc.ray_lies_on(tuple(w1), tuple(o1) + tuple(d))


# ---- TestPlane.fs -----------------------------------------------------------------------------------------------------
def test_orthogonalise_does_make_orthogonal_vectors(orc):
    """TestPlane.fs:12-26."""
    rng = np.random.default_rng(31)
    done = 0
    while done < 2000:
        o, v1, v2 = rng.normal(size=3), rng.normal(size=3), rng.normal(size=3)
        v1 /= np.linalg.norm(v1)
        v2 /= np.linalg.norm(v2)
        res = orc.plane_orthonormal_basis(tuple(o), tuple(v1), tuple(v2), (0.0, 1.0, 0.0))
        if res is None or abs(np.dot(np.cross(v1, v2), [0, 1, 0])) > 0.999:
            continue
        done += 1
        x, y = np.array(res[0]), np.array(res[1])
        assert feq(float(x @ y), 0.0) and feq(float(x @ x), 1.0) and feq(float(y @ y), 1.0)


# ---- TestPpmOutput.fs ---------------------------------------------------------------------------------------------------
def test_wikipedia_example_of_ppm_output(orc):
    """TestPpmOutput.fs:12-46 against the reference's golden file (committed as tests/golden/PpmOutputExample.txt)."""
    here = os.path.dirname(os.path.abspath(__file__))
    expected = open(os.path.join(here, "golden", "PpmOutputExample.txt"), "rb").read().replace(b"\r\n", b"\n")
    image = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255]], [[255, 255, 0], [255, 255, 255], [0, 0, 0]]], np.uint8)
    assert orc.format_ppm(image, gamma=False) == expected
