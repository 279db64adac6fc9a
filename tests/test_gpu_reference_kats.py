"""The reference's OWN unit-test vectors (RayTracing.Test/*.fs) pushed literally through the DEVICE functions (rt_dev_* hooks), and
held to the values the reference's tests expect -- not to the oracle's.  tests/test_oracle_reference_kats.py pins the oracle with the
same vectors; this file makes the device's pin direct instead of transitive ("re-expressed against the oracle's test
hooks and then against the GPU kernel").  FsCheck generators are replaced by seeded numpy generators over the same kinds of values."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DELTA = 0.00000001  # Float.tolerance (Float.fs:82)


def _ray_make(origin, vector):
    """Ray.make' (Ray.fs:26-34): direction * (1.0 / sqrt (dot direction direction)), plain IEEE doubles, left-to-right sums (Point.fs:18-35)."""
    x, y, z = (float(v) for v in vector)
    dd = (x * x + y * y) + z * z
    f = 1.0 / math.sqrt(dd)
    return [float(origin[0]), float(origin[1]), float(origin[2]), x * f, y * f, z * f]


def _lies_on_sphere(p, centre, radius):  # Sphere.liesOn (Sphere.fs:341-343): Float.equal |p - c|^2 r^2
    dx, dy, dz = p[0] - centre[0], p[1] - centre[1], p[2] - centre[2]
    return abs(((dx * dx + dy * dy) + dz * dz) - radius * radius) < DELTA


def test_intersection_of_sphere_and_ray_lies_on_both_case_1(rt):
    """TestSphereIntersection.fs:37-58, the literal case: ValueSome t, and the ray walked to t lies on the sphere."""
    ray = _ray_make((1.462205539, -4.888279676, 7.123293244), (-9.549697616, 4.400018428, 10.41024923))
    centre, radius = (-5.688391601, -5.360125644, 9.074300761), 8.199747973
    t = float(rt.hooks.sphere_first_intersection([ray], [list(centre) + [radius]])[0])
    assert not math.isnan(t)                      # ValueSome
    assert t == 12.649517791881394                # the hand evaluation of the source
    p = [ray[k] + (ray[3 + k] * t) for k in range(3)]  # Ray.walkAlong (Ray.fs:42-43)
    assert _lies_on_sphere(p, centre, radius)


def test_intersection_of_sphere_and_ray_does_lie_on_both(rt):
    """TestSphereIntersection.fs:21-34 (property): every returned intersection lies on the sphere, and none is behind the ray."""
    rng = np.random.default_rng(1)
    n = 20000
    o = rng.normal(size=(n, 3)) * 2.0
    rays = np.array([_ray_make(o[i], rng.normal(size=3)) for i in range(n)])
    sph = np.concatenate([rng.normal(size=(n, 3)) * 2, rng.normal(size=(n, 1)) * 2], axis=1)
    t = rt.hooks.sphere_first_intersection(rays, sph)
    hit = ~np.isnan(t)
    assert hit.sum() > 2000
    for i in np.nonzero(hit)[0]:
        p = [rays[i, k] + (rays[i, 3 + k] * t[i]) for k in range(3)]
        assert _lies_on_sphere(p, sph[i, :3], float(sph[i, 3])), i
    assert np.all(t[hit] > DELTA)


def _sort(x1, x2):  # TestBoundingBox.fs:13-14
    return min(x1, x2), (x1 + (DELTA / 2.0) if x1 == x2 else max(x1, x2))


@pytest.mark.parametrize("axis", [0, 1, 2])
@pytest.mark.parametrize("negate", [True, False])
def test_bounding_box_behind_the_ray_is_not_hit(rt, axis, negate):
    """TestBoundingBox.fs:16-43 (x), :45-72 (y), :86-114 (z), both signs: six of the reference's eight cases."""
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
    assert not rt.hooks.bbox_hits(rays, np.concatenate([lo, hi], axis=1)).any()


def test_bounding_box_forward_ray_going_backward_case_1(rt):
    """TestBoundingBox.fs:74-84: the degenerate zero-thickness box is not hit."""
    z1, z2 = _sort(-abs(0.0), -abs(0.0))
    x1, x2 = _sort(0.0, 0.0)
    y1, y2 = _sort(0.0, 1.0)
    assert rt.hooks.bbox_hits([[0.0, 0.0, DELTA, 0.0, 0.0, 1.0]], [[x1, y1, z1, x2, y2, z2]])[0] == 0


def test_bounding_box_forward_does_intersect_ray_going_forward(rt):
    """TestBoundingBox.fs:116-123: the unit cube is hit from inside (inverse directions +inf, products NaN)."""
    assert rt.hooks.bbox_hits([[0.0, 0.0, 0.0, 0.0, 0.0, 1.0]], [[-1.0, -1.0, -1.0, 1.0, 1.0, 1.0]])[0] == 1


def test_the_same_boxes_through_the_timed_loops_filter(rt):
    """The timed kernel's node loop runs a conservative filter above the leaves (rt_dev_bbox_filter): on the reference's own hit
    case it must say hit too; on the reference's miss cases it may say either."""
    out = rt.hooks.bbox_filter([[0.0, 0.0, 0.0, 0.0, 0.0, 1.0]], [[-1.0, -1.0, -1.0, 1.0, 1.0, 1.0]])[0]
    assert out & 1 and out & 2 and out & 4


@pytest.mark.parametrize("case", ["glass_edge", "glass_middle", "dielectric_middle"])
def test_glass_and_dielectric_known_answers(rt, case):
    """TestSphere.fs:53-84 (Glass perfectly reflects against the edge), :87-118 (Glass refracts through the middle), :121-152
    (Dielectric with probability 1.0 through the middle): ValueNone (not absorbed), colour = Colour.Green, Point.equal origin strike,
    Vector.equal direction (within 1e-8).  The reference draws from an unseeded FloatProducer; 500 seeded states stand in for it."""
    green = rt.Texture.Colour(rt.Colour.Green)
    style, centre = {
        "glass_edge": (rt.SphereStyle.Glass(1.0, green, 1.5), (0.0, 1.0, 1.0)),
        "glass_middle": (rt.SphereStyle.Glass(1.0, green, 1.5), (0.0, 0.0, 2.0)),
        "dielectric_middle": (rt.SphereStyle.Dielectric(1.0, green, 1.5, 1.0), (0.0, 0.0, 2.0)),
    }[case]
    scene = rt.Scene.make([rt.Hittable.Sphere(rt.Sphere.make(style, rt.Point.make(*centre), 1.0))])
    states = np.random.default_rng(4).integers(1, 2 ** 31 - 1, size=(500, 4), dtype=np.uint32)
    n = len(states)
    ray = np.tile([0.0, 0.0, 0.0, 0.0, 0.0, 1.0], (n, 1))       # Ray.make' origin (0, 0, 1)
    strike = np.tile([0.0, 0.0, 1.0], (n, 1))
    absorbed, col, out, _ = rt.hooks.reflection(scene, np.zeros(n, np.int32), ray, np.tile([255, 255, 255], (n, 1)), strike, states)
    assert not absorbed.any()
    assert np.all(col == [0, 255, 0])
    assert np.all(np.abs(out[:, :3] - [0.0, 0.0, 1.0]) < DELTA)
    keeps = np.ones(n, bool)
    if case == "glass_middle":
        # Through the centre Schlick's reflectionProb is R0 = ((1 - 1.5) / (1 + 1.5))^2 = 0.04 (Sphere.fs:281-292), so the reference's
        # own test fails one unseeded run in 25; here each draw decides which of the two outcomes is asserted.
        u = np.array([rt.hooks.float_producer(st, 1)[0] for st in states])
        keeps = ~(u < 0.04000000000000001)
        assert 5 < (~keeps).sum() < 45
        assert np.all(np.abs(out[~keeps, 3:] - [0.0, 0.0, -1.0]) < DELTA)
    assert np.all(np.abs(out[keeps, 3:] - [0.0, 0.0, 1.0]) < DELTA)


def test_combine_pixels_with_white_and_black(rt):
    """TestPixel.fs:157-169 (combine White = id) and :172-183 (combine Black = Black)."""
    g = np.arange(256, dtype=np.uint8)
    px = np.stack([g, g[::-1], (g.astype(np.int32) * 7 % 256).astype(np.uint8)], axis=1)
    assert np.array_equal(rt.hooks.pixel_combine(px, np.full_like(px, 255)), px)
    assert np.array_equal(rt.hooks.pixel_combine(px, np.zeros_like(px)), np.zeros_like(px))


def test_random_floats(rt):
    """TestRandom.fs:14-71: in [0, 1], normal, all ten deciles in 100 draws, consecutive draws distinct; and the state (1,2,3,4)
    known answer hand-derived from Float.fs:14-29."""
    for st in np.random.default_rng(20).integers(0, 2 ** 31 - 1, size=(30, 4), dtype=np.uint32):
        r = rt.hooks.float_producer(st, 300)
        assert np.all(np.isfinite(r)) and np.all(r >= 0.0) and np.all(r <= 1.0) and np.all(r > 2.2250738585072014e-308)
        for i in range(10):
            assert np.any((i * 0.1 < r[:100]) & (r[:100] < (i + 1) * 0.1))
        assert len(set(r[:6].tolist())) == 6
    assert rt.hooks.float_producer((1, 2, 3, 4), 4).tolist() == [0.05090332032435185, 0.1214599609657796, 0.01562500000363798, 0.12548828127921752]
