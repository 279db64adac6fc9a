"""How many boxes does a ray overlap (= node visits of the unpruned walk) in the reference's tree vs other builds?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle as orc
import ray_tracing_fsharp_amd as rt
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))

objs, cam, w, h = rt.sample_images.config3_final()
s = rt.Scene.make(objs)
skip, prim, boxes = s.tree()
o = orc.OracleScene(objs)
# sample rays: camera rays + their first two bounces, via the oracle hooks
rng = np.random.default_rng(0)
N = 20000
c = cam.to_abi()
rows = rng.integers(0, 2*h+1, N); cols = rng.integers(0, 2*w+1, N)
st = orc.stream_state(2024, (rows*(2*w+1)+cols).astype(np.uint64), rng.integers(0, 500, N).astype(np.uint32))
rays = np.zeros((N, 6))
for i in range(N):
    p = rt.FloatProducer(st[i]); r1, r2 = p.GetTwo(); st[i] = [p.x, p.y, p.z, p.w]
    lx = ((float(cols[i]-w)+r1)*c.viewport_width)/float(w); ly = ((float(h-rows[i]-1)+r2)*c.viewport_height)/float(h)
    P = np.array(c.xaxis_origin)+np.array(c.xaxis_dir)*lx+np.array(c.yaxis_dir)*ly
    d = P-np.array(c.view_origin); d /= np.sqrt(d@d); rays[i,:3]=c.view_origin; rays[i,3:]=d
allrays=[rays.copy()]
col=np.full((N,3),255,np.uint8); alive=np.arange(N); cur=rays.copy()
for gen in range(3):
    hit,strike,cnt=o.hit_object(cur[alive]); ok=hit>=0
    ab,c2,r2,g2=o.reflection(hit[ok],cur[alive][ok],col[alive][ok],strike[ok],st[alive][ok])
    idx=alive[ok]; st[idx]=g2; col[idx]=c2; cur[idx]=r2; alive=idx[ab==0]
    allrays.append(cur[alive].copy())
R=np.concatenate(allrays); print("rays", len(R))

def slab(R, lo, hi):
    with np.errstate(all="ignore"):
        inv=1.0/R[:,3:]; t0=(lo-R[:,:3])*inv; t1=(hi-R[:,:3])*inv
        sw=inv<0; a=np.where(sw,t1,t0); b=np.where(sw,t0,t1)
        tmin=np.fmax(np.fmax(a[:,0],a[:,1]),a[:,2]); tmax=np.fmin(np.fmin(b[:,0],b[:,1]),b[:,2])
        tmin=np.fmax(tmin,-np.inf); 
        return (tmax>=tmin)&(tmax>=0)

def visits(nodes):  # nodes: list of (lo,hi,children or None) in a tree; count nodes whose parent chain is all hit
    total=np.zeros(len(R)); 
    def rec(n, mask):
        nonlocal total
        total+=mask
        lo,hi,ch=n
        m=mask&slab(R,lo,hi)
        if ch:
            for k in ch: rec(k,m)
    rec(nodes, np.ones(len(R),bool)); return total

# leaf boxes of bounded spheres
A=rt._abi
leaves=[(np.array(hh.sphere.Centre)-abs(hh.sphere.Radius), np.array(hh.sphere.Centre)+abs(hh.sphere.Radius)) for hh in objs if hh.kind==A.RT_HITTABLE_SPHERE]
def ref_tree(ids):
    lo=np.min([leaves[i][0] for i in ids],0); hi=np.max([leaves[i][1] for i in ids],0)
    if len(ids)==1: return (leaves[ids[0]][0],leaves[ids[0]][1],None)
    if len(ids)==2: return (lo,hi,[ref_tree([ids[0]]),ref_tree([ids[1]])])
    best=None
    for ax in range(3):
        srt=sorted(ids,key=lambda i:leaves[i][0][ax]); half=len(srt)//2; L=srt[:half+1]; Rr=srt[half+1:]
        vol=lambda g: np.prod(np.max([leaves[i][1] for i in g],0)-np.min([leaves[i][0] for i in g],0))
        cst=vol(L)+vol(Rr)
        if best is None or cst<best[0]: best=(cst,L,Rr)
    return (lo,hi,[ref_tree(best[1]),ref_tree(best[2])])
def area(lo,hi):
    d=hi-lo; return 2*(d[0]*d[1]+d[1]*d[2]+d[0]*d[2])
def sah_tree(ids, leafcost=1.0):
    lo=np.min([leaves[i][0] for i in ids],0); hi=np.max([leaves[i][1] for i in ids],0)
    if len(ids)==1: return (leaves[ids[0]][0],leaves[ids[0]][1],None)
    best=None
    for ax in range(3):
        srt=sorted(ids,key=lambda i:(leaves[i][0][ax]+leaves[i][1][ax]))
        n=len(srt)
        # prefix/suffix boxes
        plo=[];phi=[];l=np.full(3,np.inf);hh=np.full(3,-np.inf)
        for i in srt: l=np.minimum(l,leaves[i][0]);hh=np.maximum(hh,leaves[i][1]);plo.append(l.copy());phi.append(hh.copy())
        slo=[None]*n;shi=[None]*n;l=np.full(3,np.inf);hh=np.full(3,-np.inf)
        for k in range(n-1,-1,-1): i=srt[k];l=np.minimum(l,leaves[i][0]);hh=np.maximum(hh,leaves[i][1]);slo[k]=l.copy();shi[k]=hh.copy()
        for k in range(1,n):
            cst=area(plo[k-1],phi[k-1])*(2*k-1)+area(slo[k],shi[k])*(2*(n-k)-1)
            if best is None or cst<best[0]: best=(cst,srt[:k],srt[k:])
    return (lo,hi,[sah_tree(best[1]),sah_tree(best[2])])
ids=list(range(len(leaves)))
def count(n): return 1+(sum(count(k) for k in n[2]) if n[2] else 0)
for name,t in (("reference build",ref_tree(ids)),("SAH (surface area, sweep)",sah_tree(ids))):
    v=visits(t); print(f"{name}: nodes {count(t)}, visits/ray mean {v.mean():.2f} p50 {np.percentile(v,50):.0f} p90 {np.percentile(v,90):.0f} p99 {np.percentile(v,99):.0f} max {v.max():.0f}; E[max of 64] {np.mean([rng.choice(v,64).max() for _ in range(2000)]):.1f}")
