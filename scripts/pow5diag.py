import numpy as np, sys, json
sys.path.insert(0,'.')
import ray_tracing_fsharp_amd as rt, oracle as orc
from fractions import Fraction
rng = np.random.default_rng(9)
x = np.concatenate([rng.random(1000000) * 2.0, rng.random(100000) * 1e-3, [0.0, 1.0, 2.0, 0.5, 1e-200, 1.0 - 2 ** -53]])
g = rt.hooks.arith(4, x); c = orc.arith(4, x)
bad = np.nonzero(g.view(np.uint64) != c.view(np.uint64))[0]
print("mismatches", len(bad), "of", len(x))
def cr(v):
    f = Fraction(float(v))**5
    # correctly rounded to nearest double
    return float(f)  # Fraction->float is correctly rounded in CPython
ng=nc=0
for i in bad[:2000]:
    e = cr(x[i])
    ng += (g[i]==e); nc += (c[i]==e)
print("of first", min(2000,len(bad)), "mismatches: gpu correct", ng, "glibc correct", nc)
for i in bad[:5]: print(repr(x[i]), g[i].hex(), c[i].hex(), cr(x[i]).hex())
