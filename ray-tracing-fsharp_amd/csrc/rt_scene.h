// rt_scene.h -- host side of Scene.make (Scene.fs:15-28): partition, BoundingBoxTree.make (BoundingBoxTree.fs:9-43),
// and the flattening of both into the device image described in rt_device.h.  Host-only C++ (no HIP calls here).
//
// The tree is built as index arrays straight into pre-order (node, left subtree, right subtree) with a skip link per
// node, which is the layout the stackless device walk needs; no pointer tree is ever materialised.
#pragma once
#include "../../include/rtfs_amd.h"
#include "rt_device.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <future>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

namespace rth {

struct Box { double mn[3], mx[3]; };

// BoundingBox.mergeTwo (BoundingBox.fs:96-108)
static inline Box merge_two(const Box &i, const Box &j) {
    Box o;
    for (int a = 0; a < 3; ++a) {
        o.mn[a] = i.mn[a] < j.mn[a] ? i.mn[a] : j.mn[a];
        o.mx[a] = i.mx[a] < j.mx[a] ? j.mx[a] : i.mx[a];
    }
    return o;
}
// BoundingBox.volume (BoundingBox.fs:13-16)
static inline double volume(const Box &b) { return (b.mx[0] - b.mn[0]) * (b.mx[1] - b.mn[1]) * (b.mx[2] - b.mn[2]); }

struct FlatTree {
    std::vector<int32_t> skip, prim; // prim = object-table index of a Leaf, -1 for a Branch
    std::vector<Box> box;
    std::vector<double> hits;        // probe rays that hit the box (only while a tree is being tuned)
    int depth = 0;
};

class TreeBuilder {
  public:
    TreeBuilder(const std::vector<Box> &objBoxes, FlatTree &out) : ob(objBoxes), t(out) {}

    // BoundingBoxTree.make's `go` (BoundingBoxTree.fs:14-41) over object indices `ids`
    void go(const std::vector<int32_t> &ids, int depth) {
        if (depth > t.depth) t.depth = depth;
        const size_t me = emit(merge_all(ids), -1);
        if (ids.size() == 1) { // Leaf boxes.[0]
            t.prim[me] = ids[0];
            t.box[me] = ob[(size_t) ids[0]];
        } else if (ids.size() == 2) { // Branch (Leaf boxes.[0], Leaf boxes.[1], boundAll)
            leaf(ids[0], depth + 1);
            leaf(ids[1], depth + 1);
        } else {
            // Array.sortBy on Min[axis] (BoundingBoxTree.fs:23); .NET's sort is unstable, so equal keys have no defined
            // order in the reference -- defined here as stable (DESIGN.md "Tree shape").
            struct Split { std::vector<int32_t> l, r; double cost; };
            auto split = [&](int axis) {
                Split sp;
                std::vector<int32_t> sorted = ids;
                std::stable_sort(sorted.begin(), sorted.end(),
                                 [&](int32_t a, int32_t b) { return ob[(size_t) a].mn[axis] < ob[(size_t) b].mn[axis]; });
                const size_t half = sorted.size() / 2;
                sp.l.assign(sorted.begin(), sorted.begin() + (long) half + 1); // boxes.[0 .. n/2]
                sp.r.assign(sorted.begin() + (long) half + 1, sorted.end());   // boxes.[n/2+1 ..]
                sp.cost = volume(merge_all(sp.l)) + volume(merge_all(sp.r));
                return sp;
            };
            // Large nodes: the three axes, and then the two subtrees, on threads of their own -- the same tree node for node (a subtree
            // is built into a tree of its own and appended in pre-order).  Scenes of 1e5 .. 1e6 spheres: this build was ~90 % of
            // rt_scene_create's time (1e6 spheres: 8 s of 9.5 on the GPU box's 16 cores).
            const bool threaded = depth <= kParallelDepth && ids.size() >= kParallelMin;
            Split sp[3];
            if (threaded) {
                auto f1 = std::async(std::launch::async, [&]() { return split(1); }), f2 = std::async(std::launch::async, [&]() { return split(2); });
                sp[0] = split(0);
                sp[1] = f1.get();
                sp[2] = f2.get();
            } else for (int axis = 0; axis < 3; ++axis) sp[axis] = split(axis);
            int best = 0;
            for (int axis = 1; axis < 3; ++axis) if (sp[axis].cost < sp[best].cost) best = axis; // Array.minBy: the first minimum
            if (threaded) {
                FlatTree left, right;
                auto fut = std::async(std::launch::async, [&]() { TreeBuilder b(ob, left); b.go(sp[best].l, depth + 1); });
                { TreeBuilder b(ob, right); b.go(sp[best].r, depth + 1); }
                fut.get();
                append(left);
                append(right);
            } else {
                go(sp[best].l, depth + 1);
                go(sp[best].r, depth + 1);
            }
        }
        t.skip[me] = (int32_t) t.skip.size();
    }

  private:
    const std::vector<Box> &ob;
    FlatTree &t;
    static const int kParallelDepth = 4;      // up to 16 subtrees in flight
    static const size_t kParallelMin = 4096;  // leaves below which a node is not worth a thread
    void append(const FlatTree &sub) {
        const int32_t base = (int32_t) t.skip.size();
        t.skip.reserve(t.skip.size() + sub.skip.size());
        for (size_t i = 0; i < sub.skip.size(); ++i) t.skip.push_back(sub.skip[i] + base);
        t.prim.insert(t.prim.end(), sub.prim.begin(), sub.prim.end());
        t.box.insert(t.box.end(), sub.box.begin(), sub.box.end());
        if (sub.depth > t.depth) t.depth = sub.depth;
    }
    Box merge_all(const std::vector<int32_t> &ids) const { // BoundingBox.merge = Array.reduce mergeTwo
        Box acc = ob[(size_t) ids[0]];
        for (size_t i = 1; i < ids.size(); ++i) acc = merge_two(acc, ob[(size_t) ids[i]]);
        return acc;
    }
    size_t emit(const Box &b, int32_t prim) {
        t.skip.push_back(0);
        t.prim.push_back(prim);
        t.box.push_back(b);
        return t.skip.size() - 1;
    }
    void leaf(int32_t id, int depth) {
        if (depth > t.depth) t.depth = depth;
        const size_t me = emit(ob[(size_t) id], id);
        t.skip[me] = (int32_t) t.skip.size();
    }
};

// ---- the tree the device WALKS ------------------------------------------------------------------------------------
// Scene.bestCandidate tests a sphere iff the ray hits its Leaf box and every box above it (Scene.fs:39-60).  Every Branch box
// is the exact min/max union of the Leaf boxes below it (BoundingBox.mergeTwo), and (b - o) * inv is monotonic in b even after
// rounding, so in floating point too a ray that hits a Leaf box hits every box that contains it: the set of spheres tested for
// a ray is exactly {i : hits leafBox_i}, whatever binary tree of exact unions is put above the leaves.  Among equal t^2 the
// strict `<` of Scene.fs:47 keeps the first leaf in the reference's depth-first order; with the object table sorted by that
// rank, `a < best || (a == best && object < bestObject)` reproduces it for any visiting order.  So the device may walk a
// tree built for fewer box tests -- surface-area heuristic below -- and return the same hit, bit for bit.  The reference's own
// tree (above) is still built: it defines the ranks, rt_scene_get_tree reports it, and RT_WALK_TREE_REFERENCE walks it
// (then the box-test counter equals the oracle's as well).
static inline double half_area(const Box &b) {
    double d[3];
    for (int a = 0; a < 3; ++a) { d[a] = b.mx[a] - b.mn[a]; if (!(d[a] > 0.0)) d[a] = 0.0; } // inverted (negative-radius) boxes count as empty
    return d[0] * d[1] + d[1] * d[2] + d[0] * d[2];
}
// Probe rays (rt_scene_tune): rays of a small render with the scene's camera, as the device logged them.  With them the split
// cost of a candidate box is the NUMBER OF PROBE RAYS THAT HIT IT instead of its area -- rays of this renderer start on surfaces
// inside the scene, for which area is a poor predictor -- and the finished tree is then thinned (collapse_tree below).
struct ProbeRay { double o[3], inv[3]; };
// BoundingBox.hits (BoundingBox.fs:30-94) for the tuning statistics (no result depends on it; the walk itself is rt_device.h's)
static inline bool probe_hits(const ProbeRay &r, const Box &b) {
    double tMin = -HUGE_VAL, tMax = HUGE_VAL;
    for (int a = 0; a < 3; ++a) {
        double t0 = (b.mn[a] - r.o[a]) * r.inv[a], t1 = (b.mx[a] - r.o[a]) * r.inv[a];
        if (r.inv[a] < 0.0) { const double tmp = t1; t1 = t0; t0 = tmp; }
        tMin = t0 > tMin ? t0 : tMin;
        tMax = t1 < tMax ? t1 : tMax;
        if (a < 2 && (tMax < tMin || 0.0 >= tMax)) return false;
    }
    return tMax >= tMin && tMax >= 0.0;
}

class SahBuilder {
  public:
    SahBuilder(const std::vector<Box> &objBoxes, FlatTree &out, const std::vector<ProbeRay> *probe = nullptr) : ob(objBoxes), t(out), rays(probe) {}
    void build() {
        std::vector<int32_t> ids(ob.size());
        for (size_t j = 0; j < ob.size(); ++j) ids[j] = (int32_t) j;
        suffix.resize(ob.size() + 1);
        std::vector<uint32_t> all;
        if (rays) { all.resize(rays->size()); for (size_t i = 0; i < all.size(); ++i) all[i] = (uint32_t) i; }
        go(ids.data(), ids.size(), 1, all, rays ? (double) rays->size() : 0.0, 0.0);
    }

  private:
    const std::vector<Box> &ob;
    FlatTree &t;
    const std::vector<ProbeRay> *rays;
    std::vector<double> suffix;
    std::vector<Box> pre, suf;          // boxes of ids[0, k) and ids[k, n) of the axis being swept (probe mode)
    std::vector<uint32_t> cntP, cntS;
    static const size_t kSweepMax = 4096; // above this a node is split on 32 centroid bins per axis instead of a full sweep
    static const size_t kProbeMin = 32;   // fewer probe rays than this in a box: its subtree is split by area
    static const size_t kSweepRays = 128; // rays a sweep looks at, at most (about): the split costs need a few percent only (final scene, box tests
                                          // per held-out ray after thinning: 18.08 with 512, 18.15 with 256, 18.17 with 128, 18.27 with 64; time ~linear)
    double centroid2(int32_t id, int axis) const { return ob[(size_t) id].mn[axis] + ob[(size_t) id].mx[axis]; }
    void sort_axis(int32_t *ids, size_t n, int axis) const {
        std::sort(ids, ids + n, [&](int32_t a, int32_t b) {
            const double ca = centroid2(a, axis), cb = centroid2(b, axis);
            return ca < cb || (ca == cb && a < b);
        });
    }
    // A subtree of m leaves has 2m-1 boxes, but a ray that enters it visits far fewer than all of them: m^0.8 was the best
    // exponent on the final scene (scripts/tree_collapse.py: 20.1 visits per ray with exponent 1, 18.7 with 0.8, 18.9 with 0.6).
    static double subtree_weight(size_t leaves) { return std::pow((double) (2 * leaves - 1), 0.8); }
    // `idx`: the probe rays that hit the parent's box (all of them at the root); parentHits / parentArea: the estimate this
    // node's own count is scaled from once the rays have run out
    void go(int32_t *ids, size_t n, int depth, const std::vector<uint32_t> &idx, double parentHits, double parentArea) {
        if (depth > t.depth) t.depth = depth;
        Box all = ob[(size_t) ids[0]];
        for (size_t i = 1; i < n; ++i) all = merge_two(all, ob[(size_t) ids[i]]);
        const size_t me = t.skip.size();
        t.skip.push_back(0);
        t.prim.push_back(n == 1 ? ids[0] : -1);
        t.box.push_back(all);
        std::vector<uint32_t> hidx; // the probe rays that hit this box
        double myHits = 0.0;
        if (rays) {
            if (idx.size() >= kProbeMin || depth == 1) {
                hidx.reserve(idx.size());
                for (uint32_t r : idx) if (probe_hits((*rays)[r], all)) hidx.push_back(r);
                myHits = hidx.empty() ? 1.0 : (double) hidx.size(); // a box no probe ray hit still counts as one ray's worth: the area estimates below scale from it
            } else // the rays have run out: from here down the counts are the parent's, scaled by area
                myHits = parentArea > 0.0 ? parentHits * std::min(1.0, half_area(all) / parentArea) : parentHits;
            t.hits.push_back(myHits);
        }
        const bool probe = rays && hidx.size() >= kProbeMin && n <= kSweepMax;
        if (n > 1) {
            size_t k = n / 2;
            int bestAxis = -1;
            if (depth < 48) { // a degenerate input cannot make the recursion deeper than this: below, halve by centroid order
                double bestCost = 0.0;
                for (int axis = 0; axis < 3; ++axis) {
                    if (probe) {
                        // pre[k] grows and suf[k] shrinks with k, and a ray that hits a box hits every box around it: per ray
                        // two binary searches give the first prefix and the last suffix it hits
                        sort_axis(ids, n, axis);
                        pre.resize(n + 1); suf.resize(n + 1); cntP.assign(n + 1, 0u); cntS.assign(n + 1, 0u);
                        pre[1] = ob[(size_t) ids[0]];
                        for (size_t i = 2; i < n; ++i) pre[i] = merge_two(pre[i - 1], ob[(size_t) ids[i - 1]]);
                        suf[n - 1] = ob[(size_t) ids[n - 1]];
                        for (size_t i = n - 1; i-- > 1;) suf[i] = merge_two(suf[i + 1], ob[(size_t) ids[i]]);
                        // a sweep needs the counts to a few percent only: every step-th ray of a large set (the set itself stays whole
                        // for the boxes below and for the thinning)
                        const size_t step = hidx.size() / kSweepRays > 1 ? hidx.size() / kSweepRays : 1;
                        for (size_t j = 0; j < hidx.size(); j += step) {
                            const ProbeRay &ray = (*rays)[hidx[j]];
                            size_t lo = 1, hi = n; // first k in [1, n) with pre[k] hit, n if none
                            while (lo < hi) { const size_t mid = (lo + hi) / 2; if (probe_hits(ray, pre[mid])) hi = mid; else lo = mid + 1; }
                            cntP[lo]++;
                            lo = 0; hi = n - 1; // last k in [1, n) with suf[k] hit, 0 if none
                            while (lo < hi) { const size_t mid = (lo + hi + 1) / 2; if (probe_hits(ray, suf[mid])) lo = mid; else hi = mid - 1; }
                            cntS[lo]++;
                        }
                        uint64_t hs = 0;
                        for (size_t i = n - 1; i >= 1; --i) { hs += cntS[i]; suffix[i] = (double) (hs * step); }
                        uint64_t hp = 0;
                        for (size_t i = 1; i < n; ++i) { // left = ids[0, i), right = ids[i, n)
                            hp += cntP[i];
                            const double cost = ((double) (hp * step) + 1.0) * subtree_weight(i) + (suffix[i] + 1.0) * subtree_weight(n - i);
                            if (bestAxis < 0 || cost < bestCost) { bestCost = cost; bestAxis = axis; k = i; }
                        }
                    } else if (n <= kSweepMax) {
                        sort_axis(ids, n, axis);
                        Box acc = ob[(size_t) ids[n - 1]];
                        for (size_t i = n - 1; i >= 1; --i) { if (i < n - 1) acc = merge_two(acc, ob[(size_t) ids[i]]); suffix[i] = half_area(acc); }
                        acc = ob[(size_t) ids[0]];
                        for (size_t i = 1; i < n; ++i) { // left = ids[0, i), right = ids[i, n); a subtree of m leaves has 2m-1 boxes
                            if (i > 1) acc = merge_two(acc, ob[(size_t) ids[i - 1]]);
                            const double cost = half_area(acc) * (double) (2 * i - 1) + suffix[i] * (double) (2 * (n - i) - 1);
                            if (bestAxis < 0 || cost < bestCost) { bestCost = cost; bestAxis = axis; k = i; }
                        }
                    } else {
                        const int B = 32;
                        double lo = centroid2(ids[0], axis), hi = lo;
                        for (size_t i = 1; i < n; ++i) { const double c = centroid2(ids[i], axis); if (c < lo) lo = c; if (c > hi) hi = c; }
                        if (!(hi > lo)) continue;
                        Box bb[B]; size_t cnt[B]; bool used[B];
                        for (int b = 0; b < B; ++b) { cnt[b] = 0; used[b] = false; }
                        const double scale = (double) B / (hi - lo);
                        for (size_t i = 0; i < n; ++i) {
                            int b = (int) ((centroid2(ids[i], axis) - lo) * scale); if (b >= B) b = B - 1; if (b < 0) b = 0;
                            bb[b] = used[b] ? merge_two(bb[b], ob[(size_t) ids[i]]) : ob[(size_t) ids[i]]; used[b] = true; cnt[b]++;
                        }
                        double sufA[B + 1]; size_t sufN[B + 1]; Box acc{}; bool any = false; sufN[B] = 0; sufA[B] = 0.0;
                        for (int b = B - 1; b >= 0; --b) {
                            if (used[b]) { acc = any ? merge_two(acc, bb[b]) : bb[b]; any = true; }
                            sufN[b] = sufN[b + 1] + cnt[b]; sufA[b] = any ? half_area(acc) : 0.0;
                        }
                        any = false; size_t nl = 0;
                        for (int b = 1; b < B; ++b) { // left = bins [0, b)
                            if (used[b - 1]) { acc = any ? merge_two(acc, bb[b - 1]) : bb[b - 1]; any = true; }
                            nl += cnt[b - 1];
                            if (nl == 0 || sufN[b] == 0) continue;
                            const double cost = half_area(acc) * (double) (2 * nl - 1) + sufA[b] * (double) (2 * sufN[b] - 1);
                            if (bestAxis < 0 || cost < bestCost) { bestCost = cost; bestAxis = axis; k = nl; }
                        }
                    }
                }
            }
            if (bestAxis < 0) { bestAxis = 0; k = n / 2; }
            if (n <= kSweepMax) { if (bestAxis != 2 || depth >= 48) sort_axis(ids, n, bestAxis); } // ids are in axis-2 order after the sweeps
            else std::nth_element(ids, ids + k, ids + n, [&](int32_t a, int32_t b) {
                     const double ca = centroid2(a, bestAxis), cb = centroid2(b, bestAxis);
                     return ca < cb || (ca == cb && a < b);
                 });
            const double myArea = half_area(all);
            if (rays && depth <= kParallelDepth && n >= kParallelMin) {
                // the two subtrees are independent: the left one on another thread into a tree of its own, appended in pre-order
                // afterwards -- the same tree, node for node, as the sequential build (rt_scene_tune's host time: 11.5 -> ~3 ms)
                FlatTree left, right;
                auto fut = std::async(std::launch::async, [&]() { SahBuilder b(ob, left, rays); b.suffix.resize(ob.size() + 1); b.go(ids, k, depth + 1, hidx, myHits, myArea); });
                { SahBuilder b(ob, right, rays); b.suffix.resize(ob.size() + 1); b.go(ids + k, n - k, depth + 1, hidx, myHits, myArea); }
                fut.get();
                append(left);
                append(right);
            } else {
                go(ids, k, depth + 1, hidx, myHits, myArea);
                go(ids + k, n - k, depth + 1, hidx, myHits, myArea);
            }
        }
        t.skip[me] = (int32_t) t.skip.size();
    }
    static const int kParallelDepth = 3;     // up to 8 subtrees in flight
    static const size_t kParallelMin = 48;   // leaves below which a subtree is not worth a thread
    void append(const FlatTree &sub) {
        const int32_t base = (int32_t) t.skip.size();
        for (size_t i = 0; i < sub.skip.size(); ++i) {
            t.skip.push_back(sub.skip[i] + base);
            t.prim.push_back(sub.prim[i]);
            t.box.push_back(sub.box[i]);
        }
        t.hits.insert(t.hits.end(), sub.hits.begin(), sub.hits.end());
        if (sub.depth > t.depth) t.depth = sub.depth;
    }
};

// ---- thinning the walk tree -------------------------------------------------------------------------------------------
// NOT testing a Branch box changes no result either: the walk then goes on to the Branch's children as if the box had been hit,
// and a ray that misses the box misses everything below it anyway, Leaf boxes included.  It pays where nearly every ray that
// gets as far as the Branch hits it -- one test saved for most rays, a few wasted on the children for the rest.  With H(N) = the
// number of probe rays that hit box N (a property of the box alone, by the monotonicity above) the visits of ANY choice of
// tested Branches are  sum over tested N of H(nearest tested box above N), so the best choice is a small dynamic programme over
// (node, nearest tested box above).  In the flat pre-order layout an untested Branch simply disappears: its children take its
// place among their grandparent's children, and the on-hit / on-miss links of rt_device.h need nothing new.
// On the final scene: 22.2 box tests per probe ray for the surface-area tree, 17.3 after thinning it, 15.5 for the probe-count
// build thinned (scripts/tree_collapse.py).
struct TreeThinner {
    const FlatTree &b;
    const double total; // probe rays
    std::unordered_map<uint64_t, std::pair<double, bool>> memo;
    static const size_t kExactMax = 200000; // nodes; above this the choice is made greedily, top down
    TreeThinner(const FlatTree &binary, double totalRays) : b(binary), total(totalRays) {}
    // counts below SahBuilder::kProbeMin were replaced by area-scaled estimates there, so no further smoothing is needed (the
    // small constant only keeps a box nobody hit from costing exactly nothing)
    double w(int64_t anc) const { return anc < 0 ? total + 1e-9 : b.hits[(size_t) anc] + 1e-9; }
    // expected visits inside i's subtree and whether i's box is tested, when the nearest tested box above i is `anc` (-1: none)
    std::pair<double, bool> best(int32_t i, int64_t anc) {
        if (b.prim[(size_t) i] >= 0) return {w(anc), true}; // a Leaf box is always tested: Scene.bestCandidate's own test
        if (b.skip.size() > kExactMax) return {0.0, !(w(i) >= 0.5 * w(anc))};
        const uint64_t key = ((uint64_t) (uint32_t) i << 32) | (uint64_t) (uint32_t) (anc + 1);
        auto it = memo.find(key);
        if (it != memo.end()) return it->second;
        double cT = w(anc), cS = 0.0;
        for (int32_t k = i + 1; k < b.skip[(size_t) i]; k = b.skip[(size_t) k]) { cT += best(k, i).first; cS += best(k, anc).first; }
        const std::pair<double, bool> r = cT <= cS ? std::make_pair(cT, true) : std::make_pair(cS, false);
        memo.emplace(key, r);
        return r;
    }
    void emit(int32_t i, int64_t anc, int depth, FlatTree &out) {
        const bool tested = best(i, anc).second;
        size_t me = 0;
        if (tested) {
            me = out.skip.size();
            out.skip.push_back(0); out.prim.push_back(b.prim[(size_t) i]); out.box.push_back(b.box[(size_t) i]); out.hits.push_back(b.hits[(size_t) i]);
            if (depth > out.depth) out.depth = depth;
        }
        for (int32_t k = i + 1; k < b.skip[(size_t) i]; k = b.skip[(size_t) k]) emit(k, tested ? i : anc, tested ? depth + 1 : depth, out);
        if (tested) out.skip[me] = (int32_t) out.skip.size();
    }
    // the expected box tests per probe ray of a tree as it stands (every box of it tested)
    static double visits_per_ray(const FlatTree &t, double totalRays) {
        if (t.skip.empty() || !(totalRays > 0.0)) return 0.0;
        // a node is visited by the rays that hit its parent; walk the pre-order with a stack of (end, hits)
        std::vector<std::pair<int32_t, double>> up;
        double sum = 0.0;
        for (size_t i = 0; i < t.skip.size(); ++i) {
            while (!up.empty() && (int32_t) i >= up.back().first) up.pop_back();
            sum += up.empty() ? totalRays : up.back().second;
            if (t.prim[i] < 0) up.emplace_back(t.skip[i], t.hits[i]);
        }
        return sum / totalRays;
    }
    FlatTree run() {
        FlatTree out;
        if (!b.skip.empty()) emit(0, -1, 1, out);
        return out;
    }
};

struct HostScene {
    std::vector<rt_hittable> hittables;
    std::vector<rt_texture> textures;
    std::vector<rtd::TexRec> texRecs;
    std::vector<uint8_t> texelBlob;
    std::vector<int32_t> objToOrig; // object-table index -> index in `hittables`
    std::vector<int32_t> origToObj;
    FlatTree tree;     // BoundingBoxTree.make's tree; prim = object-table index (= the leaf's depth-first rank)
    FlatTree walkTree; // what the device image holds: `tree` itself, the surface-area build over the same leaves, or a tuned tree
    int walkKind = RT_WALK_TREE_REFERENCE;
    std::vector<Box> leafBoxes; // by object-table index (= rank), for rt_scene_tune
    size_t nBounded = 0, nUnbounded = 0;
    std::vector<unsigned char> image;
    rtd::SceneOffsets off{};
};

static inline uint32_t pack_rgb(const uint8_t rgb[3]) { return (uint32_t) rgb[0] | ((uint32_t) rgb[1] << 8) | ((uint32_t) rgb[2] << 16); }

static inline size_t align16(size_t v) { return (v + 15u) & ~(size_t) 15u; }

static void encode_image(HostScene &s);
// Returns an empty string on success, otherwise the message for rt_last_error().
static std::string build_scene(const rt_hittable *h, size_t n, const rt_texture *tex, size_t ntex, int walk_kind, HostScene &s, int &status) {
    status = RT_ERR_INVALID_ARGUMENT;
    if (n > 0 && !h) return "hittables is NULL";
    if (ntex > 0 && !tex) return "textures is NULL";
    if (ntex > 254) { status = RT_ERR_UNSUPPORTED; return "at most 254 textures"; }
    s.hittables.assign(h, h + n);
    s.textures.assign(tex, tex + ntex);

    // ---- textures ----
    s.texRecs.resize(ntex);
    for (size_t i = 0; i < ntex; ++i) {
        const rt_texture &t = tex[i];
        rtd::TexRec r{};
        r.kind = t.kind;
        r.rgb = pack_rgb(t.rgb);
        r.ramp = (uint32_t) t.ramp_src[0] | ((uint32_t) t.ramp_src[1] << 8) | ((uint32_t) t.ramp_src[2] << 16);
        r.even = t.even; r.odd = t.odd;
        r.width = t.width; r.height = t.height;
        r.grid = t.grid_size;
        r.cx = t.map_centre[0]; r.cy = t.map_centre[1]; r.cz = t.map_centre[2];
        r.map_radius = t.map_radius;
        switch (t.kind) {
        case RT_TEXTURE_COLOUR: break;
        case RT_TEXTURE_UV_RAMP:
            for (int k = 0; k < 3; ++k) if (t.ramp_src[k] > RT_RAMP_V) return "texture " + std::to_string(i) + ": bad ramp source";
            break;
        case RT_TEXTURE_CHECKERED:
            if (t.even < 0 || t.odd < 0 || (size_t) t.even >= i || (size_t) t.odd >= i)
                return "texture " + std::to_string(i) + ": Checkered children must have smaller indices";
            // sin(gridSize * u), u in [0, 1]: csrc/rt_trig.h reduces arguments below 2^20 exactly
            if (!(std::fabs(t.grid_size) <= 500000.0)) { status = RT_ERR_UNSUPPORTED; return "texture " + std::to_string(i) + ": Checkered grid size beyond 5e5 (or NaN)"; }
            break;
        case RT_TEXTURE_IMAGE: {
            if (!t.texels || t.width <= 0 || t.height <= 0) return "texture " + std::to_string(i) + ": image without texels";
            const size_t bytes = (size_t) t.width * (size_t) t.height * 3u;
            r.texel_off = (uint32_t) s.texelBlob.size();
            s.texelBlob.insert(s.texelBlob.end(), t.texels, t.texels + bytes);
            while (s.texelBlob.size() % 16u) s.texelBlob.push_back(0);
            break;
        }
        default: status = RT_ERR_UNSUPPORTED; return "texture " + std::to_string(i) + ": Texture.Arbitrary closures cannot cross the C ABI";
        }
        s.texRecs[i] = r;
        s.textures[i].texels = nullptr; // not owned
    }

    // ---- Scene.make: Array.partition (snd >> ValueOption.isSome), order kept on both sides (Scene.fs:16-22) ----
    std::vector<int32_t> bounded, unbounded;
    for (size_t i = 0; i < n; ++i) {
        const rt_hittable &o = h[i];
        if (o.kind > RT_HITTABLE_INFINITE_PLANE) return "hittable " + std::to_string(i) + ": bad kind";
        const bool plane = o.kind == RT_HITTABLE_INFINITE_PLANE;
        if (plane ? o.style > RT_PLANE_FUZZED_REFLECTION : o.style > RT_SPHERE_GLASS) return "hittable " + std::to_string(i) + ": bad style";
        if (o.texture >= (int32_t) ntex) return "hittable " + std::to_string(i) + ": texture index out of range";
        if (o.texture >= 0) {
            const bool carries = plane ? o.style == RT_PLANE_LIGHT_SOURCE : o.style != RT_SPHERE_LIGHT_SOURCE_CAP;
            if (!carries) return "hittable " + std::to_string(i) + ": this style carries a Pixel, not a Texture";
        }
        (o.kind == RT_HITTABLE_SPHERE ? bounded : unbounded).push_back((int32_t) i);
    }
    const size_t nb = bounded.size(), nu = unbounded.size(), nobj = nb + nu;
    // walk offsets are int32 byte offsets with bit 30 reserved for the pending-leaf flag (RTD_LEAF)
    if (nb > 4500000u || nobj > 16000000u) { status = RT_ERR_UNSUPPORTED; return "scene too large for 32-bit walk offsets (4,500,000 bounded spheres)"; } // (2n-1) * 112 B < 2^30 (bit 30 = RTD_LEAF)
    // ---- Sphere.make's box (Sphere.fs:333-336): centre + (-r,-r,-r) .. centre + (r,r,r); inverted when r < 0 ----
    std::vector<Box> boxes(nb); // by position in `bounded` (input order), which is what BoundingBoxTree.make receives
    bool finite = true;
    for (size_t j = 0; j < nb; ++j) {
        const rt_hittable &o = h[(size_t) bounded[j]];
        for (int a = 0; a < 3; ++a) {
            boxes[j].mn[a] = o.point[a] + (-o.radius);
            boxes[j].mx[a] = o.point[a] + o.radius;
            finite = finite && std::isfinite(boxes[j].mn[a]) && std::isfinite(boxes[j].mx[a]);
        }
    }
    s.tree = FlatTree{};
    if (nb > 0) {
        std::vector<int32_t> ids(nb);
        for (size_t j = 0; j < nb; ++j) ids[j] = (int32_t) j;
        TreeBuilder(boxes, s.tree).go(ids, 1);
    }
    // Object table: bounded spheres in the order the reference's depth-first walk meets their leaves (so that an object index
    // IS the tie-breaking rank, see SahBuilder), then the unbounded list in Scene.make order.
    std::vector<int32_t> rankOf(nb, 0);
    s.objToOrig.assign(nobj, 0);
    {
        int32_t rank = 0;
        for (size_t i = 0; i < s.tree.prim.size(); ++i)
            if (s.tree.prim[i] >= 0) { const int32_t j = s.tree.prim[i]; rankOf[(size_t) j] = rank; s.objToOrig[(size_t) rank] = bounded[(size_t) j]; s.tree.prim[i] = rank; ++rank; }
    }
    for (size_t u = 0; u < nu; ++u) s.objToOrig[nb + u] = unbounded[u];
    s.origToObj.assign(n, -1);
    for (size_t j = 0; j < nobj; ++j) s.origToObj[(size_t) s.objToOrig[j]] = (int32_t) j;

    s.walkKind = (walk_kind == RT_WALK_TREE_REFERENCE || !finite || nb < 3) ? RT_WALK_TREE_REFERENCE : RT_WALK_TREE_SAH;
    s.leafBoxes.assign(nb, Box{});
    for (size_t j = 0; j < nb; ++j) s.leafBoxes[(size_t) rankOf[j]] = boxes[j];
    s.nBounded = nb; s.nUnbounded = nu;
    if (s.walkKind == RT_WALK_TREE_SAH) {
        s.walkTree = FlatTree{};
        SahBuilder(s.leafBoxes, s.walkTree).build();
    } else s.walkTree = s.tree;
    encode_image(s);
    status = RT_OK;
    return std::string();
}

// Outward rounding of a box coordinate to single precision (the node32 records of the timed kernel's filter loop, rt_device.h):
// the largest float <= v / the smallest float >= v.  +-inf and NaN pass through.
static inline float f32_down(double v) {
    float f = (float) v; // round to nearest
    if ((double) f > v) f = std::nextafterf(f, -HUGE_VALF);
    return f;
}
static inline float f32_up(double v) {
    float f = (float) v;
    if ((double) f < v) f = std::nextafterf(f, HUGE_VALF);
    return f;
}

// The device image of s.walkTree and the object table (layout: rt_device.h).  Called again when the walk tree changes.
static void encode_image(HostScene &s) {
    const rt_hittable *h = s.hittables.data();
    const size_t nb = s.nBounded, nu = s.nUnbounded, nobj = nb + nu;
    const FlatTree &wt = s.walkTree;
    const size_t nn = wt.skip.size();

    // ---- device image: node | geo | meta | node32 | mat ----
    rtd::SceneOffsets &off = s.off;
    size_t cur = 0;
    off.node = (uint32_t) cur; cur = align16(cur + nn * RTD_NODE_BYTES);
    off.geo = (uint32_t) cur;  cur = align16(cur + nobj * 48u);
    off.meta = (uint32_t) cur; cur = align16(cur + nobj * 8u);
    off.lds_total = (uint32_t) cur; // [0, lds_total): what the counting variant stages
    off.node32 = (uint32_t) cur; cur = align16(cur + nn * RTD_NODE32_BYTES);
    off.lds32_total = (uint32_t) cur - off.geo; // [geo, node32 end): what the timed variant stages
    off.mat = (uint32_t) cur;  cur = align16(cur + nobj * 32u);
    off.total = (uint32_t) cur;
    off.n_nodes = (int32_t) nn; off.n_bounded = (int32_t) nb; off.n_unbounded = (int32_t) nu;
    s.image.assign(cur == 0 ? 16 : cur, 0);
    if (cur == 0) off.total = 16;
    if (off.lds_total == 0) off.lds_total = 16;
    if (off.lds32_total == 0) off.lds32_total = 16;
    unsigned char *pnode = s.image.data() + off.node;
    unsigned char *pnode32 = s.image.data() + off.node32;
    double *pgeo = (double *) (s.image.data() + off.geo);
    int32_t *pmeta = (int32_t *) (s.image.data() + off.meta);
    double *pmat = (double *) (s.image.data() + off.mat);
    float bmax = 1e-30f;
    // node32 records are stored in order of DEPTH (breadth first; pre-order within a level): their links are explicit, so any order
    // walks the same, and this one keeps the records most often visited together -- the first part of the section is what the kernel
    // holds in LDS when the whole scene does not fit (rt_render_kernel.h, "hybrid"), and what stays in L1 / L2 otherwise.
    std::vector<uint32_t> place(nn + 1, 0); // pre-order index -> position in the node32 section; place[nn] = nn ("tree exhausted")
    {
        std::vector<uint32_t> depth(nn, 0), order(nn);
        std::vector<std::pair<int32_t, uint32_t>> up; // (end of subtree, depth) of the Branches above the current node
        for (size_t i = 0; i < nn; ++i) {
            while (!up.empty() && (int32_t) i >= up.back().first) up.pop_back();
            depth[i] = up.empty() ? 0u : up.back().second + 1u;
            if (wt.prim[i] < 0) up.emplace_back(wt.skip[i], depth[i]);
            order[i] = (uint32_t) i;
        }
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return depth[a] < depth[b]; });
        for (size_t k = 0; k < nn; ++k) place[order[k]] = (uint32_t) k;
        place[nn] = (uint32_t) nn;
    }
    for (size_t i = 0; i < nn; ++i) {
        double *bx = (double *) (pnode + i * RTD_NODE_BYTES);
        int32_t *lk = (int32_t *) (pnode + i * RTD_NODE_BYTES + 96);
        for (int a = 0; a < 3; ++a) { bx[a * 4] = wt.box[i].mn[a]; bx[a * 4 + 1] = wt.box[i].mx[a]; bx[a * 4 + 2] = wt.box[i].mx[a]; bx[a * 4 + 3] = wt.box[i].mn[a]; }
        const int32_t onMiss = wt.skip[i] * RTD_NODE_BYTES; // byte offset of the record to visit on a miss
        lk[0] = wt.prim[i] >= 0 ? (int32_t) (RTD_LEAF | (i * RTD_NODE_BYTES)) : (int32_t) ((i + 1) * RTD_NODE_BYTES); // on_hit
        lk[1] = onMiss;
        lk[2] = wt.prim[i]; // object index of a Leaf, -1 for a Branch
        lk[3] = 0;
        // the filter loop's record: the box rounded outward, {lo, hi, hi, lo} per axis; links in the queue form of node_loop_lds32:
        // a Leaf's on_hit is its on_miss (the walk goes on), the third word its queue entry, the fourth the shift that pushes it
        float *fx = (float *) (pnode32 + (size_t) place[i] * RTD_NODE32_BYTES);
        int32_t *lk32 = (int32_t *) (pnode32 + (size_t) place[i] * RTD_NODE32_BYTES + 48);
        for (int a = 0; a < 3; ++a) {
            const float lo = f32_down(wt.box[i].mn[a]), hi = f32_up(wt.box[i].mx[a]);
            fx[a * 4] = lo; fx[a * 4 + 1] = hi; fx[a * 4 + 2] = hi; fx[a * 4 + 3] = lo;
            if (std::fabs(lo) > bmax) bmax = std::fabs(lo); // (a NaN coordinate compares false and leaves bmax alone; it makes its own products NaN)
            if (std::fabs(hi) > bmax) bmax = std::fabs(hi);
        }
        const bool leaf = wt.prim[i] >= 0;
        const int32_t onMiss32 = (int32_t) (place[(size_t) wt.skip[i]] * RTD_NODE32_BYTES);
        lk32[0] = leaf ? onMiss32 : (int32_t) (place[i + 1] * RTD_NODE32_BYTES);
        lk32[1] = onMiss32;
        // the queue entry: 16 bits for scenes the LDS loop may walk (< 16384 objects), full width beyond (global loop only)
        lk32[2] = !leaf ? 0 : (nobj < 16384u ? (int32_t) (RTD_PEND_MARK | (uint32_t) wt.prim[i]) : (int32_t) (RTD_PEND_WIDE | (uint32_t) wt.prim[i]));
        lk32[3] = leaf ? 16 : 0;
    }
    off.bmax = bmax;
    {   // the hypotheses of leaf_test_object_exact's claim (rt_device.h): the exact leaf-box test is then implied by a sphere hit
        double rmax = 0.0;
        bool ok = std::isfinite(bmax) && bmax <= 1000.0f;
        for (size_t j = 0; j < nb; ++j) {
            const double r = h[(size_t) s.objToOrig[j]].radius;
            if (!(r > 0.0 && r <= 100.0)) ok = false;
            if (r > rmax) rmax = r;
        }
        off.box_implied = (ok && rmax * (double) bmax <= 500.0) ? 1 : 0;
    }
    for (size_t j = 0; j < nobj; ++j) {
        const rt_hittable &o = h[(size_t) s.objToOrig[j]];
        double *g = pgeo + j * 6;
        uint32_t m0;
        if (o.kind == RT_HITTABLE_INFINITE_PLANE) {
            g[0] = o.point[0]; g[1] = o.point[1]; g[2] = o.point[2];
            g[3] = o.normal[0]; g[4] = o.normal[1]; g[5] = o.normal[2];
            m0 = RTD_KIND_PLANE | (o.style << 2);
            pmat[j * 4 + 0] = o.albedo; pmat[j * 4 + 1] = o.fuzz; pmat[j * 4 + 2] = 0.0; pmat[j * 4 + 3] = 0.0;
        } else {
            g[0] = o.point[0]; g[1] = o.point[1]; g[2] = o.point[2];
            g[3] = o.radius * o.radius; // RadiusSquared (Sphere.fs:326)
            g[4] = o.albedo; g[5] = o.radius; // the radius itself: LightSourceCap (Sphere.fs:191) and the leaf pass' exact box (leaf_test_object_exact)
            // flipped = Float.compare this.Radius 0.0 = Less (Sphere.fs:321)
            const bool flipped = !(std::fabs(o.radius - 0.0) < 0.00000001) && (o.radius < 0.0);
            m0 = RTD_KIND_SPHERE | (o.style << 2) | (flipped ? 32u : 0u);
            const bool usesIor = o.style == RT_SPHERE_DIELECTRIC || o.style == RT_SPHERE_GLASS;
            pmat[j * 4 + 0] = 0.0; pmat[j * 4 + 1] = usesIor ? o.ior : o.fuzz; pmat[j * 4 + 2] = o.prob; pmat[j * 4 + 3] = 0.0;
            if (usesIor) {
                // Per-material values the reference recomputes at every hit from the same inputs with the same IEEE operations
                // (this translation unit is built with -ffp-contract=off): `1.0 / ior` (Sphere.fs:117, 283-284) and Schlick's
                // `param * param`, param = (1.0 - sr) / (1.0 + sr), for sr = ior (outside) and sr = 1.0 / ior (inside) (Sphere.fs:288-289).
                const double ior = o.ior;
                const double inv = 1.0 / ior;
                if (o.style == RT_SPHERE_DIELECTRIC) pmat[j * 4 + 3] = inv;
                else {
                    const double po = (1.0 - ior) / (1.0 + ior), pi = (1.0 - inv) / (1.0 + inv);
                    pmat[j * 4 + 0] = pi * pi;  // Schlick's term inside the glass
                    pmat[j * 4 + 2] = inv;      // Glass carries no refraction probability
                    pmat[j * 4 + 3] = po * po;
                }
            }
        }
        pmeta[j * 2] = (int32_t) m0;
        pmeta[j * 2 + 1] = (int32_t) (pack_rgb(o.rgb) | ((uint32_t) (o.texture + 1) << 24));
    }
}

// rt_scene_tune's host half: the walk tree rebuilt from probe rays (split costs from ray counts, then thinned) and the image
// encoded again.  Returns false (and changes nothing) for a scene that walks the reference's own tree.
struct TuneResult { double visitsBefore = 0.0, visitsAfter = 0.0; int32_t nodesBefore = 0, nodesAfter = 0; };
static bool tune_walk_tree(HostScene &s, const std::vector<ProbeRay> &rays, TuneResult &r) {
    if (s.walkKind == RT_WALK_TREE_REFERENCE || s.nBounded < 3 || rays.empty()) return false;
    const double total = (double) rays.size();
    {   // what the tree walked so far costs these rays
        const FlatTree &cur = s.walkTree;
        uint64_t visits = 0;
        for (const ProbeRay &ray : rays)
            for (size_t i = 0; i < cur.skip.size();) { ++visits; i = probe_hits(ray, cur.box[i]) ? i + 1 : (size_t) cur.skip[i]; }
        r.visitsBefore = (double) visits / total;
        r.nodesBefore = (int32_t) cur.skip.size();
    }
    FlatTree binary;
    SahBuilder(s.leafBoxes, binary, &rays).build();
    s.walkTree = TreeThinner(binary, total).run();
    s.walkKind = RT_WALK_TREE_TUNED;
    r.visitsAfter = TreeThinner::visits_per_ray(s.walkTree, total);
    r.nodesAfter = (int32_t) s.walkTree.skip.size();
    s.walkTree.hits.clear();
    encode_image(s);
    return true;
}

// ---- Camera.makeBasic (Camera.fs:34-59) with Plane.makeNormalTo' (Plane.fs:22-38) and Plane.basis (Plane.fs:82-97) ----
struct H3 { double x, y, z; };
static inline double hdot(H3 a, H3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline H3 hscale(double s, H3 v) { return H3{s * v.x, s * v.y, s * v.z}; }
static inline H3 hsum(H3 a, H3 b) { return H3{a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline bool hunit(H3 v, H3 &out) { // Vector.unitise (Point.fs:28-35)
    double d = hdot(v, v);
    if (std::fabs(d - 0.0) < 0.00000001) return false;
    double f = 1.0 / std::sqrt(d);
    out = hscale(f, v);
    return true;
}
static inline H3 hcross(H3 p, H3 q) { return H3{p.y * q.z - p.z * q.y, p.z * q.x - p.x * q.z, p.x * q.y - q.x * p.y}; } // Point.fs:44-45

static bool camera_make_basic(int32_t spp, double focal, double aspect, const double origin[3], const double dir[3], const double up[3],
                              rt_camera *out) {
    const H3 o{origin[0], origin[1], origin[2]}, v{dir[0], dir[1], dir[2]};
    const H3 corner{o.x + (v.x * focal), o.y + (v.y * focal), o.z + (v.z * focal)}; // Ray.walkAlong view focalLength
    // Plane.makeNormalTo (Plane.fs:22-36)
    H3 v1 = (std::fabs(v.z - 0.0) < 0.00000001) ? H3{0.0, 0.0, 1.0} : H3{1.0, 1.0, ((-v.x - v.y) / v.z)};
    H3 v2, v1u, upu;
    if (!hunit(hcross(v, v1), v2) || !hunit(v1, v1u)) return false; // ValueOption.get
    // Plane.basis viewUp (Plane.fs:82-97)
    if (!hunit(H3{up[0], up[1], up[2]}, upu)) return false;
    const double c1 = hdot(v1u, upu), c2 = hdot(v2, upu);
    H3 yAxis, xAxis;
    if (!hunit(hsum(hscale(c1, v1u), hscale(c2, v2)), yAxis)) return false;
    if (!hunit(hsum(hscale(c2, v1u), hscale(-c1, v2)), xAxis)) return false;
    auto st = [](double *d, H3 a) { d[0] = a.x; d[1] = a.y; d[2] = a.z; };
    st(out->view_origin, o); st(out->view_dir, v);
    st(out->xaxis_origin, corner); st(out->xaxis_dir, xAxis);
    st(out->yaxis_origin, corner); st(out->yaxis_dir, yAxis);
    out->viewport_height = 2.0;
    out->viewport_width = aspect * 2.0;
    out->focal_length = focal;
    out->samples_per_pixel = spp;
    out->bounce_depth = 150; // Camera.fs:58
    return true;
}

// ---- PixelOutput.correct (ImageOutput.fs:11-18) and ImageOutput.writePpm (ImageOutput.fs:163-197) ----
static inline uint8_t gamma_correct(uint8_t b) {
    int i = (int) std::rint(std::sqrt((double) b / 255.0) * 255.0);
    if (i == 256) i = 255;
    return (uint8_t) i;
}

static inline void append_uint(std::string &s, unsigned v) {
    char buf[12];
    int n = 0;
    do { buf[n++] = (char) ('0' + v % 10u); v /= 10u; } while (v);
    while (n) s.push_back(buf[--n]);
}

static std::string format_ppm(const uint8_t *rgb, int32_t rows, int32_t cols, bool gamma) {
    uint8_t lut[256];
    for (int i = 0; i < 256; ++i) lut[i] = gamma ? gamma_correct((uint8_t) i) : (uint8_t) i;
    std::string s;
    s.reserve((size_t) rows * (size_t) cols * 12u + 32u);
    s += "P3\n";
    append_uint(s, (unsigned) cols); s.push_back(' '); append_uint(s, (unsigned) rows); s.push_back('\n');
    s += "255\n";
    for (int32_t r = 0; r < rows; ++r) {
        for (int32_t c = 0; c < cols; ++c) {
            const uint8_t *p = rgb + ((size_t) r * (size_t) cols + (size_t) c) * 3u;
            append_uint(s, lut[p[0]]); s.push_back(' ');
            append_uint(s, lut[p[1]]); s.push_back(' ');
            append_uint(s, lut[p[2]]);
            if (c != cols - 1) s.push_back(' ');
        }
        if (r != rows - 1) s.push_back('\n'); // no trailing newline (ImageOutput.fs:191-196)
    }
    return s;
}


// ---- ImageOutput.resume's temp-file format (ImageOutput.fs:131-161) and readPixelMap (ImageOutput.fs:46-113) ----
// Per pixel, row-major: writeAsciiInt row ; ',' ; writeAsciiInt col ; '\n' ; R G B as three raw bytes.
// writeAsciiInt emits NOTHING for 0 (ImageOutput.fs:115-129), which consumeAsciiInteger reads back as 0.
static inline void append_ascii_int_ref(std::string &s, int v) { // the reference's digit loop, quirk included
    int tmp = v, pow = 1;
    while (tmp > 0) { tmp /= 10; pow *= 10; }
    pow /= 10;
    while (pow > 0) { s.push_back((char) ('0' + (v / pow) % 10)); pow /= 10; }
}
static std::string format_pixel_map(const uint8_t *rgb, int32_t rows, int32_t cols) {
    std::string s;
    s.reserve((size_t) rows * (size_t) cols * 12u);
    for (int32_t r = 0; r < rows; ++r)
        for (int32_t c = 0; c < cols; ++c) {
            const uint8_t *p = rgb + ((size_t) r * (size_t) cols + (size_t) c) * 3u;
            append_ascii_int_ref(s, r); s.push_back(',');
            append_ascii_int_ref(s, c); s.push_back('\n');
            s.push_back((char) p[0]); s.push_back((char) p[1]); s.push_back((char) p[2]);
        }
    return s;
}
// Returns the number of pixels set, or -1 when a (row, col) lies outside the image (the reference would throw).
// A truncated tail is ignored exactly as the reference's `go` does (ImageOutput.fs:69-106).
static int64_t parse_pixel_map(const uint8_t *data, size_t n, int32_t rows, int32_t cols, uint8_t *rgb, uint8_t *present) {
    size_t pos = 0;
    auto consume = [&](int &out) -> bool { // consumeAsciiInteger: digits, then ONE terminating non-digit; EOF -> ValueNone
        int answer = 0;
        while (pos < n) {
            const int ch = data[pos++];
            if (48 <= ch && ch <= 57) answer = 10 * answer + (ch - 48);
            else { out = answer; return true; }
        }
        return false;
    };
    int64_t count = 0;
    for (;;) {
        int row, col;
        if (!consume(row) || !consume(col)) break;
        if (pos + 3 > n) break;
        if (row < 0 || row >= rows || col < 0 || col >= cols) return -1;
        const size_t o = (size_t) row * (size_t) cols + (size_t) col;
        rgb[o * 3] = data[pos]; rgb[o * 3 + 1] = data[pos + 1]; rgb[o * 3 + 2] = data[pos + 2];
        pos += 3;
        if (present) present[o] = 1;
        ++count;
    }
    return count;
}

} // namespace rth
