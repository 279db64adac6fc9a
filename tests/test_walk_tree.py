"""The tree the device walks (rt_scene.h SahBuilder), host side only: whatever its shape, it must be a binary tree over exactly
the reference's Leaf boxes whose every Branch box is the exact union of the leaves below it -- that, and the rank tie-break, is
all the "same hits bit for bit" argument of DESIGN.md "Walk tree" needs.  No GPU involved."""
import numpy as np
import pytest

import scenes

rt = scenes.rt
P, S, H, Px, Tex = scenes.P, scenes.S, scenes.H, scenes.Px, scenes.Tex


def _spheres(n, seed, negative=False, flat=False):
    rng = np.random.default_rng(seed)
    objs = []
    for i in range(n):
        c = P(float(rng.uniform(-9, 9)), 0.2 if flat else float(rng.uniform(0, 4)), float(rng.uniform(-3, 15)))
        r = float(rng.uniform(0.05, 0.4)) * (-1.0 if negative and i % 7 == 0 else 1.0)
        objs.append(H.Sphere(rt.Sphere.make(S.LambertReflection(0.5, Tex(Px(9, 9, 9))), c, r)))
    objs.insert(n // 2, H.UnboundedSphere(rt.Sphere.make(S.LightSource(Tex(Px(1, 2, 3))), P(0.0, 0.0, 0.0), 500.0)))
    return objs


def _check(skip, prim, boxes):
    n = len(skip)
    leaves = []

    def rec(i):  # returns (next index, union box of the leaves below i)
        if prim[i] >= 0:
            assert skip[i] == i + 1
            leaves.append(int(prim[i]))
            return i + 1, boxes[i].copy()
        j, left = rec(i + 1)
        k, right = rec(j)
        assert skip[i] == k
        u = np.array([min(left[0], right[0]), max(left[1], right[1]), min(left[2], right[2]), max(left[3], right[3]),
                      min(left[4], right[4]), max(left[5], right[5])])
        assert np.array_equal(u.view(np.uint64), boxes[i].view(np.uint64)), f"node {i}: box is not the exact union of its leaves"
        return k, u

    import sys
    sys.setrecursionlimit(20000)
    end, _ = rec(0)
    assert end == n
    return leaves


def _inner_area(skip, prim, boxes):
    d = np.maximum(boxes[:, 1::2] - boxes[:, 0::2], 0.0)
    a = d[:, 0] * d[:, 1] + d[:, 1] * d[:, 2] + d[:, 0] * d[:, 2]
    return float(a[prim < 0].sum())


@pytest.mark.parametrize("n,kw", [(3, {}), (4, {}), (57, {}), (485, {"flat": True}), (700, {"negative": True}), (9000, {})])
def test_sah_walk_tree_is_a_tree_of_exact_unions_over_the_same_leaves(n, kw):
    objs = _spheres(n, seed=n, **kw)
    s = rt.Scene.make(objs)
    info = s.info()
    assert info["walk_tree"] == 0 and info["n_nodes"] == 2 * n - 1
    ref = s.tree()
    walk = s.walk_tree()
    ref_leaves, walk_leaves = _check(*ref), _check(*walk)
    assert sorted(ref_leaves) == sorted(walk_leaves) and len(set(walk_leaves)) == n
    # a leaf carries the same sphere's box in both trees
    box_of = {int(p): ref[2][i] for i, p in enumerate(ref[1]) if p >= 0}
    for i, p in enumerate(walk[1]):
        if p >= 0:
            assert np.array_equal(walk[2][i].view(np.uint64), box_of[int(p)].view(np.uint64))
    if n >= 57 and not kw.get("negative"):  # expected box tests per ray ~ summed area of the Branch boxes
        assert _inner_area(*walk) < _inner_area(*ref)
    assert info["walk_tree_depth"] <= 48 + int(np.ceil(np.log2(n))) + 1


def test_reference_walk_tree_is_the_reference_tree():
    objs = _spheres(200, seed=1)
    rt.set_walk_tree("reference")
    try:
        s = rt.Scene.make(objs)
    finally:
        rt.set_walk_tree("sah")
    assert s.info()["walk_tree"] == 1
    for a, b in zip(s.tree(), s.walk_tree()):
        assert np.array_equal(a, b)


def test_degenerate_inputs_fall_back_or_stay_shallow():
    # all spheres coincident: every split costs the same; non-finite coordinates: the reference's tree is walked
    same = [H.Sphere(rt.Sphere.make(S.LambertReflection(0.5, Tex(Px(9, 9, 9))), P(1.0, 1.0, 1.0), 0.5)) for _ in range(300)]
    s = rt.Scene.make(same)
    assert sorted(_check(*s.walk_tree())) == sorted(_check(*s.tree()))
    assert s.info()["walk_tree_depth"] <= 60
    bad = _spheres(20, seed=3)
    bad[4] = H.Sphere(rt.Sphere.make(S.LambertReflection(0.5, Tex(Px(9, 9, 9))), P(float("inf"), 0.0, 0.0), 0.5))
    assert rt.Scene.make(bad).info()["walk_tree"] == 1


def test_a_ray_that_hits_a_box_hits_every_box_containing_it(orc):
    """The monotonicity the walk-tree argument rests on, checked on the oracle's BoundingBox.hits (the literal restatement of
    BoundingBox.fs:30-94): 3 M (ray, inner box, outer box) triples with outer = exact union of inner and another box, including
    axis-parallel rays (infinite inverse directions), origins exactly on faces (0 * inf = NaN paths), origins inside the
    boxes, shared faces and inverted (negative-radius) boxes.  hits(inner) must imply hits(outer)."""
    rng = np.random.default_rng(123)
    n = 3_000_000
    rays = scenes.random_rays(n, 5, origin_scale=3.0)
    k = n // 8
    rays[:k, 3:] = np.eye(3)[rng.integers(0, 3, k)] * rng.choice([-1.0, 1.0], (k, 1))            # axis-parallel
    two = rng.integers(0, 3, k)
    rays[k:2 * k, 3:][np.arange(k), two] = 0.0                                                    # one zero component
    rays[k:2 * k, 3:] /= np.linalg.norm(rays[k:2 * k, 3:], axis=1, keepdims=True)
    lo = rng.normal(size=(n, 3)) * 2.0
    inner = np.concatenate([lo, lo + rng.uniform(0.0, 2.0, (n, 3))], axis=1)
    inner[2 * k:2 * k + 1000, 3:] = inner[2 * k:2 * k + 1000, :3] - 0.3                           # inverted boxes
    olo = rng.normal(size=(n, 3)) * 2.0
    other = np.concatenate([olo, olo + rng.uniform(0.0, 2.0, (n, 3))], axis=1)
    aim = (inner[:, :3] + inner[:, 3:]) / 2.0 + rng.normal(size=(n, 3)) * 0.7 - rays[:, :3]       # half the rays aim at the inner box
    aim /= np.linalg.norm(aim, axis=1, keepdims=True)
    sel = rng.random(n) < 0.5
    sel[:2 * k] = False
    rays[sel, 3:] = aim[sel]
    on_face = slice(3 * k, 4 * k)
    rays[on_face, 0] = inner[on_face, 0]                                                          # origin exactly on the min-x face
    rays[4 * k:5 * k, :3] = (inner[4 * k:5 * k, :3] + inner[4 * k:5 * k, 3:]) / 2.0               # origin inside
    outer = np.concatenate([np.minimum(inner[:, :3], other[:, :3]), np.maximum(inner[:, 3:], other[:, 3:])], axis=1)
    hi, ho = orc.bbox_hits(rays, inner), orc.bbox_hits(rays, outer)
    assert 0.15 < hi.mean() < 0.85
    assert not np.any((hi == 1) & (ho == 0))
