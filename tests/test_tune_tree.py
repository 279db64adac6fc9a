"""rt_scene_tune's host half (rt_scene_tune_rays: csrc/rt_scene.h SahBuilder with probe rays + TreeThinner), no GPU involved.  A tuned
walk tree is n-ary, and the "same hits bit for bit" argument needs exactly what it needs of the binary one: the same Leaf boxes,
each once, and every Branch box the exact union of the leaves below it (then a ray that hits a Leaf box hits every box above it,
tested or not -- test_walk_tree.py checks that monotonicity on the oracle's BoundingBox.hits)."""
import numpy as np
import pytest

import scenes

rt = scenes.rt
P, S, H, Px, Tex = scenes.P, scenes.S, scenes.H, scenes.Px, scenes.Tex


def check_nary(skip, prim, boxes):
    """Pre-order / skip-link consistency, at least two children per Branch, exact unions; returns the leaves' objects in walk order."""
    n = len(skip)
    leaves = []

    def rec(i):
        if prim[i] >= 0:
            assert skip[i] == i + 1
            leaves.append(int(prim[i]))
            return boxes[i].copy()
        u, k, kids = None, i + 1, 0
        while k < skip[i]:
            b = rec(k)
            u = b if u is None else np.array([min(u[0], b[0]), max(u[1], b[1]), min(u[2], b[2]), max(u[3], b[3]), min(u[4], b[4]), max(u[5], b[5])])
            k = int(skip[k]); kids += 1
        assert k == skip[i] and kids >= 2, f"node {i}: {kids} children, subtree ends at {k}, skip link {skip[i]}"
        assert np.array_equal(u.view(np.uint64), boxes[i].view(np.uint64)), f"node {i}: box is not the exact union of its leaves"
        return u

    import sys
    sys.setrecursionlimit(20000)
    k = 0
    while k < n:  # an untested root leaves a forest
        assert 0 < skip[k] <= n
        rec(k)
        k = int(skip[k])
    assert k == n
    return leaves


def visits(skip, prim, boxes, rays):
    """Box tests per ray of the skip-link walk (numpy slab test: statistics only)."""
    with np.errstate(all="ignore"):
        inv = 1.0 / rays[:, 3:]
    total = 0
    at = np.zeros(len(rays), np.int64)
    live = np.arange(len(rays))
    n = len(skip)
    while len(live):
        total += len(live)
        i = at[live]
        b = boxes[i]
        o = rays[live, :3]
        with np.errstate(all="ignore"):
            t0 = (b[:, 0::2] - o) * inv[live]; t1 = (b[:, 1::2] - o) * inv[live]
        lo = np.where(inv[live] < 0, t1, t0); hi = np.where(inv[live] < 0, t0, t1)
        tmin = np.fmax(np.fmax(lo[:, 0], lo[:, 1]), lo[:, 2]); tmax = np.fmin(np.fmin(hi[:, 0], hi[:, 1]), hi[:, 2])
        hit = (tmax >= tmin) & (tmax >= 0)
        at[live] = np.where(hit, i + 1, skip[i])
        live = live[at[live] < n]
    return total / len(rays)


def probe_rays(orc, objs, cam, w, h, n=3000, seed=5, bounces=8):
    """Rays of real paths (camera rays and their bounces) through the oracle's hooks -- what rt_scene_tune's probe logs on the GPU."""
    rng = np.random.default_rng(seed)
    c = cam.to_abi()
    rows = rng.integers(0, 2 * h + 1, n); cols = rng.integers(0, 2 * w + 1, n)
    st = orc.stream_state(seed, (rows * (2 * w + 1) + cols).astype(np.uint64), rng.integers(0, 50, n).astype(np.uint32))
    rays = np.zeros((n, 6))
    for i in range(n):
        p = rt.FloatProducer(st[i]); r1, r2 = p.GetTwo(); st[i] = [p.x, p.y, p.z, p.w]
        lx = ((float(cols[i] - w) + r1) * c.viewport_width) / float(w); ly = ((float(h - rows[i] - 1) + r2) * c.viewport_height) / float(h)
        pt = np.array(c.xaxis_origin) + np.array(c.xaxis_dir) * lx + np.array(c.yaxis_dir) * ly
        d = pt - np.array(c.view_origin); d /= np.sqrt(d @ d)
        rays[i, :3] = c.view_origin; rays[i, 3:] = d
    out = [rays.copy()]
    o = orc.OracleScene(objs)
    col = np.full((n, 3), 255, np.uint8); alive = np.arange(n); cur = rays.copy()
    for _ in range(bounces):
        if len(alive) == 0:
            break
        hit, strike, _cnt = o.hit_object(cur[alive]); ok = hit >= 0
        ab, c2, r2, g2 = o.reflection(hit[ok], cur[alive][ok], col[alive][ok], strike[ok], st[alive][ok])
        idx = alive[ok]; st[idx] = g2; col[idx] = c2; cur[idx] = r2; alive = idx[ab == 0]
        out.append(cur[alive].copy())
    return np.concatenate(out)


def test_tuned_tree_of_the_final_scene(orc):
    objs, cam, w, h = rt.sample_images.config3_final()
    rays = probe_rays(orc, objs, cam, w, h)
    rng = np.random.default_rng(0)
    rays = rays[rng.permutation(len(rays))]
    train, test = rays[: len(rays) // 2], rays[len(rays) // 2:]
    s = rt.Scene.make(objs)
    before = s.walk_tree()
    info = s.tune_rays(train)
    after = s.walk_tree()
    assert info["tuned"] == 1 and info["probe_rays"] == len(train) and info["probe_rows"] == 0
    assert s.info()["walk_tree"] == rt._abi.RT_WALK_TREE_TUNED and s.info()["n_nodes"] == len(before[0]) and s.info()["walk_tree_nodes"] == len(after[0]) == info["nodes_after"] < info["nodes_before"] == len(before[0])
    leaves = check_nary(*after)
    assert sorted(leaves) == sorted(int(p) for p in before[1] if p >= 0) and len(set(leaves)) == s.info()["n_bounded"]
    box_of = {int(p): before[2][i] for i, p in enumerate(before[1]) if p >= 0}
    for i, p in enumerate(after[1]):
        if p >= 0:
            assert np.array_equal(after[2][i].view(np.uint64), box_of[int(p)].view(np.uint64))
    # the library's own estimate is of the rays it was given; on rays it has not seen the gain must hold up
    vb, va = visits(*before, test), visits(*after, test)
    assert abs(info["box_tests_before"] - visits(*before, train)) < 0.2 and abs(info["box_tests_after"] - visits(*after, train)) < 0.2
    assert va < 0.8 * vb, (vb, va)
    # the same rays give the same tree; tuning a tuned scene starts from the leaves again
    t = rt.Scene.make(objs); t.tune_rays(train)
    assert all(np.array_equal(a, b) for a, b in zip(t.walk_tree(), after))
    s.tune_rays(train)
    assert all(np.array_equal(a, b) for a, b in zip(s.walk_tree(), after))
    # the reference's own tree is still reported as it was
    assert all(np.array_equal(a, b) for a, b in zip(s.tree(), rt.Scene.make(objs).tree()))


def test_scenes_that_are_left_alone(orc):
    objs, cam, w, h = scenes.small_final()
    rays = scenes.random_rays(500, 3)
    s = rt.Scene.make(objs, walk_tree="reference")
    before = s.walk_tree()
    info = s.tune_rays(rays)
    assert info["tuned"] == 0 and info["nodes_before"] == info["nodes_after"] == len(before[0]) and s.info()["walk_tree"] == 1
    assert all(np.array_equal(a, b) for a, b in zip(s.walk_tree(), before))
    s = rt.Scene.make(objs)
    assert s.tune_rays(np.zeros((0, 6)))["tuned"] == 0 and s.info()["walk_tree"] == 0
    two = [o for o in objs if o.kind != rt._abi.RT_HITTABLE_SPHERE] + [o for o in objs if o.kind == rt._abi.RT_HITTABLE_SPHERE][:2]
    s = rt.Scene.make(two)
    assert s.tune_rays(rays)["tuned"] == 0
    with pytest.raises(Exception):
        rt.check(rt.lib.rt_scene_tune_rays(None, None, 0, None))


def test_a_probe_that_misses_the_scene_leaves_a_sane_tree():
    """No probe ray hits anything: the counts run out at the root and every box below is priced by area -- the result must still be
    a hierarchy (the area model thins a little), not a flat list of leaves under the root."""
    objs, cam, w, h = rt.sample_images.config3_final()
    s = rt.Scene.make(objs)
    rays = np.zeros((1000, 6)); rays[:, 1] = 100.0; rays[:, 4] = 1.0
    info = s.tune_rays(rays)
    skip, prim, boxes = s.walk_tree()
    check_nary(skip, prim, boxes)
    assert info["tuned"] == 1 and skip[0] == len(skip)  # the root is tested: nothing of the probe gets past it
    assert 0.8 * info["nodes_before"] < info["nodes_after"] < info["nodes_before"] and s.info()["walk_tree_depth"] >= 6
    assert info["box_tests_before"] == 1.0


@pytest.mark.parametrize("seed", range(12))
def test_any_rays_give_a_valid_tree(seed):
    """Whatever the probe looks like -- few rays, rays that miss everything, axis-parallel and zero directions, NaNs -- the result is a
    tree of exact unions over the same leaves."""
    rng = np.random.default_rng(seed)
    n = int(rng.integers(3, 400))
    objs = []
    for i in range(n):
        c = P(float(rng.uniform(-9, 9)), float(rng.uniform(0, 4)), float(rng.uniform(-3, 15)))
        r = float(rng.uniform(0.05, 0.9)) * (-1.0 if i % 5 == 0 and seed % 2 else 1.0)
        objs.append(H.Sphere(rt.Sphere.make(S.LambertReflection(0.5, Tex(Px(9, 9, 9))), c, r)))
        if i % 9 == 0:
            objs.append(objs[-1])  # exact duplicates
    m = int(rng.choice([1, 7, 40, 3000, 40000]))
    rays = scenes.random_rays(m, seed, origin_scale=float(rng.choice([0.1, 5.0, 100.0])))
    if seed % 3 == 0:
        rays[::3, 3 + seed % 3] = 0.0  # axis-parallel
        rays[1::50, 3:] = 0.0
        rays[2::50, 0] = np.nan
    s = rt.Scene.make(objs)
    before = s.walk_tree()
    info = s.tune_rays(rays)
    assert info["tuned"] == 1
    after = s.walk_tree()
    leaves = check_nary(*after)
    assert sorted(leaves) == sorted(int(p) for p in before[1] if p >= 0)


def test_the_threaded_build_is_deterministic(orc):
    """rt_scene_tune's host build puts the two subtrees of the upper Branches on separate threads and appends them in pre-order: the
    tree must be the same, node for node, every time (the contract says "the same call yields the same tree"), whatever the threads'
    timing -- ten builds of the final scene from the same rays."""
    objs, cam, w, h = rt.sample_images.config3_final()
    rays = probe_rays(orc, objs, cam, w, h, n=1500)
    first = None
    for _ in range(10):
        s = rt.Scene.make(objs)
        assert s.tune_rays(rays)["tuned"] == 1
        t = s.walk_tree()
        if first is None:
            first = t
        else:
            assert all(np.array_equal(a, b) for a, b in zip(first, t))
