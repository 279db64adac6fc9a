"""Distribution of tree-node visits per ray by bounce generation, for the bench scene (CPU, oracle hooks)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle as orc
import ray_tracing_fsharp_amd as rt

objs, cam, w, h = rt.sample_images.config3_final()
o = orc.OracleScene(objs)
c = cam.to_abi()
rng = np.random.default_rng(0)
N = 40000
rows = rng.integers(0, 2*h+1, N); cols = rng.integers(0, 2*w+1, N)
pix = (rows * (2*w+1) + cols).astype(np.uint64)
st = orc.stream_state(2024, pix, rng.integers(0, 500, N).astype(np.uint32))
# camera rays (Scene.traceOnce): two draws from each stream
rays = np.zeros((N, 6))
for i in range(N):
    p = rt.FloatProducer(st[i]); r1, r2 = p.GetTwo(); st[i] = [p.x, p.y, p.z, p.w]
    lx = ((float(cols[i] - w) + r1) * c.viewport_width) / float(w)
    ly = ((float(h - rows[i] - 1) + r2) * c.viewport_height) / float(h)
    P = np.array(c.xaxis_origin) + np.array(c.xaxis_dir) * lx + np.array(c.yaxis_dir) * ly
    d = P - np.array(c.view_origin); d = d / np.sqrt(d @ d)
    rays[i, :3] = c.view_origin; rays[i, 3:] = d
col = np.full((N, 3), 255, np.uint8)
alive = np.arange(N)
allc = []
for gen in range(8):
    hit, strike, cnt = o.hit_object(rays[alive])
    a = cnt[:, 0].astype(np.int64)
    allc.append(a)
    print(f"gen {gen}: rays {len(alive):6d}  aabb mean {a.mean():6.1f}  p50 {np.percentile(a,50):5.0f}  p90 {np.percentile(a,90):5.0f}  p99 {np.percentile(a,99):5.0f} max {a.max():4d}  prim mean {cnt[:,1].mean():.2f}")
    ok = hit >= 0
    ab, c2, r2, g2 = o.reflection(hit[ok], rays[alive][ok], col[alive][ok], strike[ok], st[alive][ok])
    idx = alive[ok]
    st[idx] = g2; col[idx] = c2; rays[idx] = r2
    alive = idx[ab == 0]
    if len(alive) == 0: break
allc = np.concatenate(allc)
print("all rays: mean", allc.mean(), "E[max of 64 random]", np.mean([rng.choice(allc, 64).max() for _ in range(2000)]))
sec = np.concatenate([x for x in [allc[N:]]])
print("secondary only: mean", sec.mean(), "E[max of 64]", np.mean([rng.choice(sec, 64).max() for _ in range(2000)]))
pr = allc[:N]
print("primary only: mean", pr.mean(), "E[max of 64 random]", np.mean([rng.choice(pr, 64).max() for _ in range(2000)]))
