"""Would the walk be shorter if some Branch boxes were NOT tested?  Skipping a Branch's test never changes a result (a ray
that misses a box misses every box inside it), it only replaces one test by the children's; it pays where almost every ray that
reaches the Branch hits it.  Here: the surface-area tree of the final scene, rays sampled as in tree_quality.py, a top-down
choice per Branch on one half of the rays, the visit counts on the other half."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle as orc  # noqa: E402
import ray_tracing_fsharp_amd as rt  # noqa: E402

objs, cam, w, h = rt.sample_images.config3_final()
o = orc.OracleScene(objs)
rng = np.random.default_rng(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
c = cam.to_abi()
rows = rng.integers(0, 2 * h + 1, N); cols = rng.integers(0, 2 * w + 1, N)
st = orc.stream_state(2024, (rows * (2 * w + 1) + cols).astype(np.uint64), rng.integers(0, 500, N).astype(np.uint32))
rays = np.zeros((N, 6))
for i in range(N):
    p = rt.FloatProducer(st[i]); r1, r2 = p.GetTwo(); st[i] = [p.x, p.y, p.z, p.w]
    lx = ((float(cols[i] - w) + r1) * c.viewport_width) / float(w); ly = ((float(h - rows[i] - 1) + r2) * c.viewport_height) / float(h)
    P = np.array(c.xaxis_origin) + np.array(c.xaxis_dir) * lx + np.array(c.yaxis_dir) * ly
    d = P - np.array(c.view_origin); d /= np.sqrt(d @ d); rays[i, :3] = c.view_origin; rays[i, 3:] = d
allrays = [rays.copy()]
col = np.full((N, 3), 255, np.uint8); alive = np.arange(N); cur = rays.copy()
for gen in range(50):
    if len(alive) == 0:
        break
    hit, strike, cnt = o.hit_object(cur[alive]); ok = hit >= 0
    ab, c2, r2, g2 = o.reflection(hit[ok], cur[alive][ok], col[alive][ok], strike[ok], st[alive][ok])
    idx = alive[ok]; st[idx] = g2; col[idx] = c2; cur[idx] = r2; alive = idx[ab == 0]
    allrays.append(cur[alive].copy())
R = np.concatenate(allrays)
perm = rng.permutation(len(R)); R = R[perm]
print("rays", len(R), [len(a) for a in allrays])
with np.errstate(all="ignore"):
    INV = 1.0 / R[:, 3:]
SW = INV < 0


def slab(lo, hi, m):
    with np.errstate(all="ignore"):
        O = R[m, :3]; inv = INV[m]; sw = SW[m]
        t0 = (lo - O) * inv; t1 = (hi - O) * inv
        a = np.where(sw, t1, t0); b = np.where(sw, t0, t1)
        tmin = np.fmax(np.fmax(a[:, 0], a[:, 1]), a[:, 2]); tmax = np.fmin(np.fmin(b[:, 0], b[:, 1]), b[:, 2])
        return (tmax >= tmin) & (tmax >= 0)


s = rt.Scene.make(objs)
skip, prim, boxes = s.walk_tree() if hasattr(s, "walk_tree") else s.tree()
skip = np.asarray(skip); prim = np.asarray(prim); boxes = np.asarray(boxes)
print("walk tree nodes", len(skip))


def children(i):
    if prim[i] >= 0:
        return []
    out = []; k = i + 1
    while k < skip[i]:
        out.append(k); k = skip[k]
    return out


half = len(R) // 2
train = np.arange(half); test = np.arange(half, len(R))


def cost(i, idx, tested):
    """visits below and including i for the rays idx, given the set of tested nodes"""
    if i in tested or prim[i] >= 0:
        n = len(idx)
        idx = idx[slab(boxes[i, 0::2], boxes[i, 1::2], idx)]
    else:
        n = 0
    return n + sum(cost(k, idx, tested) for k in children(i)) if len(idx) else n


def choose(i, idx, tested):
    """top-down: decide for Branch i on the rays idx; returns the cost"""
    if prim[i] >= 0:
        tested.add(i)
        return len(idx)
    if len(idx) == 0:
        mark_all(i, tested)
        return 0
    hitm = slab(boxes[i, 0::2], boxes[i, 1::2], idx)
    p = hitm.mean()
    tA = set(); cA = len(idx) + sum(choose(k, idx[hitm], tA) for k in children(i)); tA.add(i)
    if p > float(os.environ.get("PMIN", "0.45")):
        tB = set(); cB = sum(choose(k, idx, tB) for k in children(i))
        if cB < cA:
            tested |= tB
            return cB
    tested |= tA
    return cA


def mark_all(i, tested):
    tested.add(i)
    for k in children(i):
        mark_all(k, tested)


sys.setrecursionlimit(10000)
everything = set(range(len(skip)))
base_tr = cost(0, train, everything); base_te = cost(0, test, everything)
tested = set()
c_tr = choose(0, train, tested)
c_te = cost(0, test, tested)
nb = int((prim < 0).sum())
print(f"all Branches tested: {base_tr / len(train):.2f} (train) {base_te / len(test):.2f} (test) visits per ray")
print(f"{nb - len([i for i in tested if prim[i] < 0])} of {nb} Branch tests skipped: {c_tr / len(train):.2f} (train) {c_te / len(test):.2f} (test)")
# by generation (the permutation is undone through `gen_of`)
gen_of = np.concatenate([np.full(len(a), g) for g, a in enumerate(allrays)])[perm]
for name, sel in (("camera rays", gen_of == 0), ("bounced rays", gen_of > 0)):
    idx = test[sel[test]]
    print(f"  {name}: {cost(0, idx, everything) / len(idx):.2f} -> {cost(0, idx, tested) / len(idx):.2f}")
tr2 = train[gen_of[train] > 0]
tested2 = set(); choose(0, tr2, tested2)
idx = test[gen_of[test] > 0]
print(f"chosen on bounced rays only: bounced {cost(0, idx, everything) / len(idx):.2f} -> {cost(0, idx, tested2) / len(idx):.2f}; ", end="")
idx = test[gen_of[test] == 0]
print(f"camera {cost(0, idx, everything) / len(idx):.2f} -> {cost(0, idx, tested2) / len(idx):.2f}; skipped {nb - len([i for i in tested2 if prim[i] < 0])}")


# ---- the same choice from box areas alone (what a scene builder can do without rays) ----
def harea(i):
    d = boxes[i, 1::2] - boxes[i, 0::2]
    return d[0] * d[1] + d[1] * d[2] + d[0] * d[2]


from functools import lru_cache  # noqa: E402


def area_choice(gamma):
    ar = [harea(i) for i in range(len(skip))]
    ch = [children(i) for i in range(len(skip))]

    @lru_cache(maxsize=None)
    def f(i, anc):  # expected visits in the subtree of i per ray that hit the tested ancestor `anc` (-1: none, every ray)
        if prim[i] >= 0:
            return 1.0, True
        p = 1.0 if anc < 0 else min(1.0, (ar[i] / ar[anc]) ** gamma)
        cT = 1.0 + p * sum(f(k, i)[0] for k in ch[i]) if anc >= 0 else 1.0 + sum(f(k, i)[0] for k in ch[i])
        cS = sum(f(k, anc)[0] for k in ch[i])
        return (cT, True) if cT <= cS else (cS, False)

    out = set()

    def mark(i, anc):
        c, t = f(i, anc)
        if t:
            out.add(i)
        for k in ch[i]:
            mark(k, i if t else anc)
    mark(0, -1)
    return out


idxB = test[gen_of[test] > 0]; idxC = test[gen_of[test] == 0]
for gamma in (1.0, 0.75, 0.5, 0.35, 0.25):
    t = area_choice(gamma)
    print(f"area model, p = (area ratio)^{gamma}: skipped {nb - len([i for i in t if prim[i] < 0])}; bounced {cost(0, idxB, t) / len(idxB):.2f}; camera {cost(0, idxC, t) / len(idxC):.2f}")


# ---- from per-node hit counts H(N) of a probe (what the device can count): a ray that hits a box hits every enclosing box, so
# H(N) is the number of probe rays that hit N whatever the tree above it looks like, and the visits of ANY choice are
# sum over tested N of H(nearest tested ancestor of N) ----
def hit_counts(idx):
    H = np.zeros(len(skip))

    def rec(i, ix):
        ix = ix[slab(boxes[i, 0::2], boxes[i, 1::2], ix)]
        H[i] = len(ix)
        if len(ix):
            for k in children(i):
                rec(k, ix)
    rec(0, idx)
    return H


def weight_choice(W, total, greedy=None):
    ch = [children(i) for i in range(len(skip))]

    @lru_cache(maxsize=None)
    def F(i, a):
        if prim[i] >= 0:
            return a, True
        cT = a + sum(F(k, W[i])[0] for k in ch[i])
        cS = sum(F(k, a)[0] for k in ch[i])
        return (cT, True) if cT <= cS else (cS, False)

    out = set()

    def mark(i, a):
        if greedy is None:
            t = F(i, a)[1]
        else:
            t = prim[i] >= 0 or not (W[i] >= greedy * a)
        if t:
            out.add(i)
        for k in ch[i]:
            mark(k, W[i] if t else a)
    mark(0, total)
    return out


for frac in (1.0, 0.1, 0.01):
    sub = train[: max(50, int(len(train) * frac))]
    H = hit_counts(sub) + 1.0
    t = weight_choice(H, float(len(sub)))
    print(f"hit counts of {len(sub)} probe rays, exact choice: skipped {nb - len([i for i in t if prim[i] < 0])}; test rays {cost(0, test, t) / len(test):.2f}")
H = hit_counts(train) + 1.0
for g in (0.4, 0.5, 0.6, 0.7, 0.8):
    t = weight_choice(H, float(len(train)), greedy=g)
    print(f"  greedy, skip when H(N) >= {g} * H(tested ancestor): skipped {nb - len([i for i in t if prim[i] < 0])}; test rays {cost(0, test, t) / len(test):.2f}")
