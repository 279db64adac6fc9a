"""(ray, box) pairs built to sit where a conservative single-precision box filter is most likely to go wrong, and a numpy model
of that filter (csrc/rt_device.h, "the node loop of the timed variant").  Shared by tests/test_filter_conservative.py (CPU model
against the oracle's exact BoundingBox.hits) and tests/test_gpu_parity.py (the device's filter against the device's exact test).

The property under test is ONE implication, for every pair: BoundingBox.hits (BoundingBox.fs:30-94) says hit  =>  the filter says hit.
"""
import numpy as np

F32 = np.float32


def _unit(v):
    n = np.linalg.norm(v, axis=1, keepdims=True)
    return v / np.where(n == 0, 1.0, n)


def classes(n, seed):
    """Yields (name, rays[n,6], boxes[n,6]) -- boxes as (min xyz, max xyz).  Each class keeps its coordinates within one order of
    magnitude or so, so that the filter's margin scale (the largest |coordinate| of the batch) is as tight as it is for a real tree."""
    rng = np.random.default_rng(seed)

    def boxes_at(scale, size):
        lo = rng.normal(size=(n, 3)) * scale
        return np.concatenate([lo, lo + np.abs(rng.normal(size=(n, 3))) * size], axis=1)

    def rays_through(points, dist_scale):
        d = _unit(rng.normal(size=(n, 3)))
        t = np.exp(rng.uniform(np.log(1e-3), np.log(dist_scale), n)) * rng.choice([1.0, -1.0], n, p=[0.85, 0.15])
        return np.concatenate([points - d * t[:, None], d], axis=1)

    def on_boundary(b, kind):
        """points on faces (kind 1), edges (2) or corners (3) of the boxes: `kind` coordinates pinned to lo or hi"""
        u = rng.random((n, 3))
        p = b[:, :3] + (b[:, 3:] - b[:, :3]) * u
        pin = np.argsort(rng.random((n, 3)), axis=1) < kind  # which axes are pinned
        side = rng.random((n, 3)) < 0.5
        pinned = np.where(side, b[:, :3], b[:, 3:])
        return np.where(pin, pinned, p)

    for scale, size, dist in ((2.0, 1.0, 30.0), (10.0, 0.4, 30.0), (1000.0, 2.0, 3000.0), (1.0, 1e-4, 10.0), (1e6, 1e3, 1e7), (1e-3, 1e-3, 1.0)):
        b = boxes_at(scale, size)
        tag = f"scale{scale:g}_size{size:g}"
        # ordinary rays
        r = np.concatenate([rng.normal(size=(n, 3)) * scale, _unit(rng.normal(size=(n, 3)))], axis=1)
        yield f"random_{tag}", r, b
        # rays through points ON faces, edges, corners: the exact decision hangs on the last bits
        for kind in (1, 2, 3):
            p = on_boundary(b, kind)
            r = rays_through(p, dist)
            # a third of them nudged by a few ulps either way
            nudge = (rng.integers(-3, 4, size=(n, 3)) * (rng.random((n, 1)) < 0.34)).astype(np.float64)
            r[:, :3] += nudge * np.spacing(np.abs(r[:, :3]) + 1e-300)
            yield f"boundary{kind}_{tag}", r, b
        # origins exactly on a face / corner, or inside
        r = np.concatenate([on_boundary(b, 1), _unit(rng.normal(size=(n, 3)))], axis=1)
        yield f"origin_on_face_{tag}", r, b
        r = np.concatenate([np.where(rng.random((n, 3)) < 0.5, b[:, :3], b[:, 3:]), _unit(rng.normal(size=(n, 3)))], axis=1)
        yield f"origin_on_corner_{tag}", r, b
        # axis-aligned and nearly axis-aligned directions: +-0.0, denormal-in-float and denormal-in-double components
        d = _unit(rng.normal(size=(n, 3)))
        tiny = rng.choice([0.0, -0.0, 1e-320, -1e-320, 1e-300, 1e-60, -1e-45, 1e-40, -1e-38, 1e-30, 1e-20, -1e-12], size=(n, 3))
        mask = rng.random((n, 3)) < 0.5
        mask[mask.all(axis=1), 0] = False
        d = np.where(mask, tiny, d)
        nz = np.linalg.norm(d, axis=1, keepdims=True)
        d = np.where(mask, d, d / nz)  # keep the tiny components as they are
        o = on_boundary(b, 2) - d * rng.uniform(0.0, dist, (n, 1))
        o = np.where(rng.random((n, 1)) < 0.3, on_boundary(b, 1), o)  # origins on a face: 0 * inf = NaN products
        yield f"axis_aligned_{tag}", np.concatenate([o, d], axis=1), b
        # inverted (negative-radius spheres, Sphere.fs:333-336) and degenerate boxes
        bi = b.copy()
        sel = rng.random(n) < 0.5
        bi[sel, :3], bi[sel, 3:] = b[sel, 3:], b[sel, :3]
        flat = rng.integers(0, 3, n)
        bi[~sel, 3 + flat[~sel]] = bi[~sel, flat[~sel]]  # zero thickness on one axis (TestBoundingBox.fs:74-84)
        yield f"inverted_or_flat_{tag}", rays_through(on_boundary(b, 1), dist), bi
    # far-away origins and directions that are not unit vectors (the exact test does not care; neither may the filter)
    b = boxes_at(5.0, 1.0)
    r = rays_through(on_boundary(b, 2), 1e13)
    yield "far_origins", r, b
    r = rays_through(on_boundary(b, 2), 30.0)
    r[:, 3:] *= np.exp(rng.uniform(np.log(1e-30), np.log(1e30), (n, 1)))
    yield "non_unit_directions", r, b
    # non-finite rays and boxes: the exact test ignores NaN products; with everything NaN it says "hit"
    r = rays_through(on_boundary(b, 1), 30.0)
    bad = rng.choice([np.nan, np.inf, -np.inf, 1e308, -1e308, 1e200], size=(n, 6))
    where = rng.random((n, 6)) < 0.25
    yield "non_finite_rays", np.where(where, bad, r), b
    b2 = np.where(rng.random((n, 6)) < 0.2, rng.choice([np.nan, np.inf, -np.inf, 1e300, -1e300], size=(n, 6)), b)
    yield "non_finite_boxes", r, b2


def f32_down(v):
    f = v.astype(F32)
    up = f.astype(np.float64) > v
    return np.where(up, np.nextafter(f, F32(-np.inf)), f)


def f32_up(v):
    f = v.astype(F32)
    dn = f.astype(np.float64) < v
    return np.where(dn, np.nextafter(f, F32(np.inf)), f)


def _fma32(a, b, c):
    # a * b is exact in double (24 + 24 bits); the sum is rounded to double and then to float: within an ulp of the fused result
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(F32)


def model(rays, boxes, bmax=0.0, rcp_ulps=0):
    """The filter in numpy.  rcp_ulps in {-1, 0, +1}: v_rcp_f32 is within one ulp of the reciprocal; the model takes the correctly
    rounded value moved by that many ulps, so the three runs bracket whatever the hardware returns."""
    with np.errstate(all="ignore"):
        o, d = rays[:, :3], rays[:, 3:]
        lo, hi = f32_down(boxes[:, :3]), f32_up(boxes[:, 3:])
        bm = F32(1e-30)
        for a in (lo, hi):
            fin = np.abs(a[~np.isnan(a)])
            if fin.size:
                bm = max(bm, F32(fin.max()))
        if bmax > float(bm):
            bm = f32_up(np.array([bmax]))[0]
        d32 = d.astype(F32)
        inv = (1.0 / d32.astype(np.float64)).astype(F32)
        if rcp_ulps:
            inv = np.nextafter(inv, F32(np.inf) if rcp_ulps > 0 else F32(-np.inf))
        oi = o.astype(F32) * inv
        m = _fma32(np.abs(inv), np.full_like(inv, bm), np.abs(oi)) * F32(2.0 ** -20) + F32(1e-30)
        cn, cf = -oi - m, m - oi
        neg = np.signbit(d)
        near, far = np.where(neg, hi, lo), np.where(neg, lo, hi)
        tn, tf = _fma32(near, inv, cn), _fma32(far, inv, cf)
        tmin = np.fmax(np.fmax(np.fmax(tn[:, 0], tn[:, 1]), tn[:, 2]), F32(0.0))  # fmax/fmin return the non-NaN operand, as v_max3/v_min3 do
        tmax = np.fmin(np.fmin(tf[:, 0], tf[:, 1]), tf[:, 2])
        return ~(tmax < tmin)
