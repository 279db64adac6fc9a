"""Back-of-envelope simulation of wave scheduling policies for the render kernel (costs in VALU instructions per wave)."""
import numpy as np, sys
rng = np.random.default_rng(1)
# empirical-ish node visit distribution: lognormal fitted to mean 26, p50 25, p90 38, p99 62, tail to 150
def sample_nodes(n):
    x = rng.lognormal(np.log(23.5), 0.42, n)
    tail = rng.random(n) < 0.01
    x[tail] *= rng.uniform(1.5, 3.0, tail.sum())
    return np.maximum(1, x.astype(int))
C_NODE, C_LEAF, C_SHADE, C_REFILL = 31, 60, 700, 300
P_END = 0.31
ITEMS = 16 * 489

class Ray:
    __slots__ = ("nodes", "leaves")
def new_ray():
    n = int(sample_nodes(1)[0]); k = rng.poisson(1.3)
    pos = set(rng.integers(0, n, k).tolist()) if k else set()
    return [n, pos, 0]  # total nodes, leaf positions, progress

def sim_current():
    cost = 0; rays = 0; nxt = 0
    lanes = [None] * 64
    while True:
        need = [i for i in range(64) if lanes[i] is None]
        if need and nxt < ITEMS:
            for i in need:
                if nxt < ITEMS: lanes[i] = new_ray(); nxt += 1
            cost += C_REFILL
        act = [i for i in range(64) if lanes[i] is not None]
        if not act: break
        # traversal with deferred leaves: all lanes until done
        while True:
            running = [i for i in act if lanes[i][2] < lanes[i][0]]
            if not running: break
            pend = set()
            cur = list(running)
            while cur:
                cost += C_NODE
                nxtcur = []
                for i in cur:
                    r = lanes[i]; p = r[2]; r[2] += 1
                    if p in r[1]: pend.add(i)
                    elif r[2] < r[0]: nxtcur.append(i)
                cur = nxtcur
            if pend: cost += C_LEAF
        cost += C_SHADE; rays += len(act)
        for i in act:
            lanes[i] = None if rng.random() < P_END else new_ray()
    return cost / rays

def sim_threshold(W, RMIN):
    cost = 0; rays = 0; nxt = 0
    lanes = [None] * 64   # None idle; else ray list + state: r[3]: 0 trav,1 pending leaf,2 done
    def mk():
        r = new_ray(); r.append(0); return r
    while True:
        idle = [i for i in range(64) if lanes[i] is None]
        busy = 64 - len(idle)
        if idle and nxt < ITEMS and (len(idle) >= RMIN or busy == 0):
            for i in idle:
                if nxt < ITEMS: lanes[i] = mk(); nxt += 1
            cost += C_REFILL
        if all(l is None for l in lanes): break
        # node loop until no traversers or waiting >= W
        while True:
            trav = [i for i in range(64) if lanes[i] is not None and lanes[i][3] == 0]
            waiting = sum(1 for l in lanes if l is not None and l[3] != 0)
            if not trav or waiting >= W: break
            cost += C_NODE
            for i in trav:
                r = lanes[i]; p = r[2]; r[2] += 1
                if p in r[1]: r[3] = 1
                elif r[2] >= r[0]: r[3] = 2
        pend = [i for i in range(64) if lanes[i] is not None and lanes[i][3] == 1]
        if pend:
            cost += C_LEAF
            for i in pend:
                r = lanes[i]; r[3] = 0 if r[2] < r[0] else 2
        done = [i for i in range(64) if lanes[i] is not None and lanes[i][3] == 2]
        trav = [i for i in range(64) if lanes[i] is not None and lanes[i][3] == 0]
        if done and (len(done) >= W or not trav):
            cost += C_SHADE; rays += len(done)
            for i in done:
                lanes[i] = None if rng.random() < P_END else mk()
    return cost / rays

ideal = (26 * C_NODE + 1.3 * C_LEAF + C_SHADE + C_REFILL * P_END) / 64
print("ideal per ray", ideal)
c = sim_current(); print("current policy", c, "util", ideal / c)
for W in (16, 24, 32, 40, 48):
    for R in (1, 8, 16):
        c = sim_threshold(W, R); print(f"W={W} RMIN={R}", round(c, 1), "util", round(ideal / c, 3))

def sim_greedy(wn=1.0, wl=1.0, ws=1.0, wr=1.0, batch_nodes=1):
    """Each step run the stage with the largest weighted lane count."""
    cost = 0; rays = 0; nxt = 0
    lanes = [None] * 64
    def mk():
        r = new_ray(); r.append(0); return r
    while True:
        idle = [i for i in range(64) if lanes[i] is None]
        trav = [i for i in range(64) if lanes[i] is not None and lanes[i][3] == 0]
        pend = [i for i in range(64) if lanes[i] is not None and lanes[i][3] == 1]
        done = [i for i in range(64) if lanes[i] is not None and lanes[i][3] == 2]
        nidle = len(idle) if nxt < ITEMS else 0
        if not trav and not pend and not done and nidle == 0: break
        scores = {"n": wn * len(trav), "l": wl * len(pend), "s": ws * len(done), "r": wr * nidle}
        pick = max(scores, key=scores.get)
        if pick == "n":
            for _ in range(batch_nodes):
                trav = [i for i in range(64) if lanes[i] is not None and lanes[i][3] == 0]
                if not trav: break
                cost += C_NODE
                for i in trav:
                    r = lanes[i]; p = r[2]; r[2] += 1
                    if p in r[1]: r[3] = 1
                    elif r[2] >= r[0]: r[3] = 2
        elif pick == "l":
            cost += C_LEAF
            for i in pend:
                r = lanes[i]; r[3] = 0 if r[2] < r[0] else 2
        elif pick == "s":
            cost += C_SHADE; rays += len(done)
            for i in done:
                lanes[i] = None if rng.random() < P_END else mk()
        else:
            for i in idle:
                if nxt < ITEMS: lanes[i] = mk(); nxt += 1
            cost += C_REFILL
    return cost / rays

print("greedy equal weights", ideal / sim_greedy())
for wn, wl, ws, wr in ((1, 1, 1, 1), (1.5, 1, 1, 1), (2, 1, 1, 1), (1, 2, 1, 1), (1, 1, 1.5, 1), (1, 1, 1, 0.5), (1.5, 1.5, 1, 0.7), (2, 2, 1, 0.7), (3, 3, 1, 1), (2, 3, 1, 0.5)):
    print((wn, wl, ws, wr), round(ideal / sim_greedy(wn, wl, ws, wr), 3))

def sim_two_thresholds(L, Y, RMIN):
    """node loop exits when pending >= L or done >= Y or nobody is walking; leaf stage if any pending (>= L or stalled);
    shade when done >= Y or nobody walking."""
    cost = 0; rays = 0; nxt = 0
    lanes = [None] * 64
    def mk():
        r = new_ray(); r.append(0); return r
    while True:
        idle = [i for i in range(64) if lanes[i] is None]
        busy = 64 - len(idle)
        if idle and nxt < ITEMS and (len(idle) >= RMIN or busy == 0):
            for i in idle:
                if nxt < ITEMS: lanes[i] = mk(); nxt += 1
            cost += C_REFILL
        if all(l is None for l in lanes): break
        while True:
            trav = [i for i in range(64) if lanes[i] is not None and lanes[i][3] == 0]
            npend = sum(1 for l in lanes if l is not None and l[3] == 1)
            ndone = sum(1 for l in lanes if l is not None and l[3] == 2)
            if not trav or npend >= L or ndone >= Y: break
            cost += C_NODE
            for i in trav:
                r = lanes[i]; p = r[2]; r[2] += 1
                if p in r[1]: r[3] = 1
                elif r[2] >= r[0]: r[3] = 2
        pend = [i for i in range(64) if lanes[i] is not None and lanes[i][3] == 1]
        trav = [i for i in range(64) if lanes[i] is not None and lanes[i][3] == 0]
        done = [i for i in range(64) if lanes[i] is not None and lanes[i][3] == 2]
        if pend and (len(pend) >= L or not trav or len(done) >= Y):
            cost += C_LEAF
            for i in pend:
                r = lanes[i]; r[3] = 0 if r[2] < r[0] else 2
        done = [i for i in range(64) if lanes[i] is not None and lanes[i][3] == 2]
        trav = [i for i in range(64) if lanes[i] is not None and lanes[i][3] == 0]
        pend = [i for i in range(64) if lanes[i] is not None and lanes[i][3] == 1]
        if done and (len(done) >= Y or (not trav and not pend)):
            cost += C_SHADE; rays += len(done)
            for i in done:
                lanes[i] = None if rng.random() < P_END else mk()
    return cost / rays

print("--- calibrated costs: node 30, leaf 70, shade 700, refill 250 ---")
C_NODE, C_LEAF, C_SHADE, C_REFILL = 30, 70, 700, 250
ideal = (26 * C_NODE + 1.3 * C_LEAF + C_SHADE + C_REFILL * P_END) / 64
print("one threshold W=44 R=12:", round(ideal / sim_threshold(44, 12), 3))
for L in (8, 16, 24, 32):
    for Y in (24, 32, 40, 48):
        print(f"L={L} Y={Y}", round(ideal / sim_two_thresholds(L, Y, 12), 3))
