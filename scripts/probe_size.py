"""How much does rt_scene_tune's tree depend on the number of probe rays?  CPU only: rays of the bench scene from the oracle (camera rays
and their first bounces), a tree tuned with n of them (rt_scene_tune_rays), box visits per ray of a held-out set walked in numpy.
usage: python scripts/probe_size.py [n ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import oracle as orc  # noqa: E402
import ray_tracing_fsharp_amd as rt  # noqa: E402

objs, cam, w, h = rt.sample_images.config3_final()
o = orc.OracleScene(objs)
c = cam.to_abi()


def rays_of(n, seed):
    rng = np.random.default_rng(seed)
    rows = rng.integers(0, 2 * h + 1, n); cols = rng.integers(0, 2 * w + 1, n)
    st = orc.stream_state(2024, (rows * (2 * w + 1) + cols).astype(np.uint64), rng.integers(0, 500, n).astype(np.uint32))
    rays = np.zeros((n, 6))
    vo, xo, xd, yd = (np.array(v) for v in (c.view_origin, c.xaxis_origin, c.xaxis_dir, c.yaxis_dir))
    for i in range(n):
        p = rt.FloatProducer(st[i]); r1, r2 = p.GetTwo(); st[i] = [p.x, p.y, p.z, p.w]
        lx = ((float(cols[i] - w) + r1) * c.viewport_width) / float(w); ly = ((float(h - rows[i] - 1) + r2) * c.viewport_height) / float(h)
        d = xo + xd * lx + yd * ly - vo
        rays[i, :3] = vo; rays[i, 3:] = d / np.sqrt(d @ d)
    out = [rays.copy()]
    col = np.full((n, 3), 255, np.uint8); alive = np.arange(n); cur = rays.copy()
    for _ in range(6):
        hit, strike, _cnt = o.hit_object(cur[alive]); ok = hit >= 0
        ab, c2, r2, g2 = o.reflection(hit[ok], cur[alive][ok], col[alive][ok], strike[ok], st[alive][ok])
        idx = alive[ok]; st[idx] = g2; col[idx] = c2; cur[idx] = r2; alive = idx[ab == 0]
        if len(alive) == 0:
            break
        out.append(cur[alive].copy())
    return np.concatenate(out)


def visits(scene, R):
    skip, prim, boxes = scene.walk_tree()
    n = len(skip)
    with np.errstate(all="ignore"):
        inv = 1.0 / R[:, 3:]
    pos = np.zeros(len(R), np.int64); total = np.zeros(len(R)); leaves = np.zeros(len(R))
    live = np.arange(len(R))
    while len(live):
        i = pos[live]
        b = boxes[i]
        with np.errstate(all="ignore"):
            t0 = (b[:, 0::2] - R[live, :3]) * inv[live]; t1 = (b[:, 1::2] - R[live, :3]) * inv[live]
        sw = inv[live] < 0
        a = np.where(sw, t1, t0); f = np.where(sw, t0, t1)
        tmin = np.fmax(np.fmax(a[:, 0], a[:, 1]), a[:, 2]); tmax = np.fmin(np.fmin(f[:, 0], f[:, 1]), f[:, 2])
        hit = (tmax >= tmin) & (tmax >= 0)
        total[live] += 1
        leaves[live] += hit & (prim[i] >= 0)
        pos[live] = np.where(hit, i + 1, skip[i])
        live = live[pos[live] < n]
    return total.mean(), leaves.mean()


test = rays_of(6000, 99)
print("held-out rays:", len(test), flush=True)
s = rt.Scene.make(objs)
print("as built: visits/ray %.2f, leaf boxes hit %.2f" % visits(s, test), flush=True)
pool = rays_of(60000, 7)
np.random.default_rng(1).shuffle(pool)
for n in [int(x) for x in sys.argv[1:]] or [2000, 7000, 20000, 60000, len(pool)]:
    s = rt.Scene.make(objs)
    info = s.tune_rays(pool[:n])
    v, l = visits(s, test)
    print(f"probe {n:7d} rays: nodes {info['nodes_after']}, visits/ray {v:.2f}, leaf boxes hit {l:.2f}, build {info['build_ms']:.1f} ms", flush=True)
