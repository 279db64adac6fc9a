// rt_render_kernel.h -- the persistent render megakernel for gfx950.
//
// Takes the role of Scene.render / renderPixel / traceOnce / traceRay (Scene.fs:93-236) for one shard of image rows.
//
// Execution model:
//   * One workgroup per CU-slot, scene image staged ONCE per workgroup into LDS (<= ~146 KiB at 1024 threads and 16-pixel units), then every
//     WAVE is an independent worker pulling work units (runs of pixels) from a global queue, one atomic per unit.
//   * Inside a unit the 64 lanes are path slots in one of three states (idle / walking the tree / walk finished); stages
//     (refill, node loop, leaf tests, shade) run when enough lanes want them -- see run_items.
//   * Adaptive sampling (Scene.fs:172-194): phase 1 traces 2k+1 samples of every pixel, splitting the byte sums after
//     sample k; the wave compares the two integer means per pixel and ballot-compacts the pixels that must continue;
//     phase 2 traces their remaining spp-2k-1 samples -- either right away on the same wave (fused mode) or in a second
//     launch over the surviving pixels ordered longest-job-first (two-pass mode, for small shards) -- see render_kernel.
//   * Per-sample colours are summed with LDS atomics into the wave's own accumulator words; each pixel's PixelStats
//     {Count,SumRed,SumGreen,SumBlue} is written as one 16-byte store per lane (coalesced), plus the mean RGB.
//   No cross-workgroup communication exists inside a launch, so no fences are needed; queues and counters are relaxed
//   device atomics, and pass B only reads what the stream-ordered earlier launches wrote.
#pragma once
#include "rt_device.h"

namespace rtd {

#define RTD_MAX_CHUNK 64

struct RenderParams {
    const CameraParams *cam_ptr;   // device copy of the camera: read where a camera ray is built, not held in SGPRs
    int32_t depth;                 // Camera.BounceDepth
    int32_t spp;                   // Camera.SamplesPerPixel
    int32_t max_w, max_h;
    SceneOffsets off;
    const unsigned char *scene_image; // global copy of the image described by `off`
    const TexRec *tex;
    const uint8_t *texels;
    uint64_t seed_key;
    int32_t cols;                  // 2*max_w+1
    int32_t row_first, row_stride; // image rows of this shard: row_first + i*row_stride
    int32_t n_rows;
    int32_t k;                     // firstTrial = min 5 (spp/2)   (Scene.fs:172)
    int32_t chunk;                 // pixels per work unit, <= 64
    int32_t yield_lanes;           // see run_items: a stage yields once this many lanes wait for another stage
    int32_t refill_lanes;          // see run_items: idle lanes are refilled once this many are idle
    int32_t *accum;                // [n_rows*cols][4]
    uint8_t *rgb;                  // [n_rows*cols][3] or null
    unsigned long long *counters;  // [16]: rays, aabb, prim, refl, samples, pixels_early, -, -, then stage executions (COUNT):
                                   //       [8] refill [9] node trips [10] leaf stages [11] shade stages [12] lanes refilled [13] lanes shaded
    unsigned int *queue;           // work-unit counter of the fused kernel / of pass A, zeroed before launch
    // two-pass rendering (see render_kernel): pass A appends (cost << 32 | local pixel) for every pixel that continues;
    // pass B walks `live_list` (those pixels ordered by decreasing cost) with its own queue
    unsigned long long *pairs;
    const unsigned int *live_list;
    unsigned int *live_count;      // number of pairs / list entries (device side)
    unsigned long long *queue_b;   // next unassigned list entry
    uint32_t total_waves;          // waves of the grid (guided unit sizes in pass B)
};

// Per-wave LDS scratch (in 4-byte words), P = pixels per work unit:
//   acc  [2][P][3]  sums of slot 0 / slot 1
//   pix  [P][4]     row, col, pixel stream key lo, hi
//   live [P]        compacted pixel slots for phase 2
//   cost [P]        rays traced for the pixel in phase 1 (pass A only: the cost estimate that orders pass B)
#define RTD_WAVE_WORDS(P) (14u * (uint32_t) (P)) /* pass B uses it as two slots of {acc [P][3], pix [P][4]} */

// Wave-private LDS words: adds from many lanes may land on one word (same pixel), so they are ds_add_u32; the owner
// lane later takes the sum and clears the word in one ds_wrxchg.  One wave's LDS operations execute in order.
RTD_INLINE void lds_add(RTD_AS3 uint32_t *p, uint32_t v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
RTD_INLINE uint32_t lds_take(RTD_AS3 uint32_t *p) { return __hip_atomic_exchange(p, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }

RTD_INLINE uint64_t wave_sum_u64(uint64_t v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor((unsigned long long) v, off, 64);
    return v;
}

template <bool LDS> RTD_INLINE SceneView<LDS> make_view(const RenderParams &p, const unsigned char *lds_base);
template <> RTD_INLINE SceneView<true> make_view<true>(const RenderParams &p, const unsigned char *lds_base) {
    SceneView<true> v;
    const RTD_AS3 unsigned char *b = (const RTD_AS3 unsigned char *) lds_base;
    v.node = (Ptrs<true>::bp) (b + p.off.node);
    v.geo = (Ptrs<true>::d2p) (b + p.off.geo);
    v.meta = (Ptrs<true>::i2p) (b + p.off.meta);
    v.mat = (Ptrs<true>::dp) (b + p.off.mat);
    v.n_nodes = p.off.n_nodes; v.n_bounded = p.off.n_bounded; v.n_unbounded = p.off.n_unbounded;
    v.first = (int) (uint32_t) (uintptr_t) v.node; v.end = v.first + v.n_nodes * RTD_NODE_BYTES; // links were made absolute at staging
    v.tex = p.tex; v.texels = p.texels;
    return v;
}
template <> RTD_INLINE SceneView<false> make_view<false>(const RenderParams &p, const unsigned char *) {
    SceneView<false> v;
    const unsigned char *b = p.scene_image;
    v.node = b + p.off.node;
    v.geo = (const d2 *) (b + p.off.geo);
    v.meta = (const i2 *) (b + p.off.meta);
    v.mat = (const double *) (b + p.off.mat);
    v.n_nodes = p.off.n_nodes; v.n_bounded = p.off.n_bounded; v.n_unbounded = p.off.n_unbounded;
    v.first = 0; v.end = v.n_nodes * RTD_NODE_BYTES;
    v.tex = p.tex; v.texels = p.texels;
    return v;
}

// Scheduling thresholds of run_items (lanes of a wave):
//   yield_lanes  the node loop yields once this many lanes are waiting for another stage (a pending leaf test, or a finished
//              walk that wants shading); the shade stage runs once this many walks are finished (or none is left walking)
//   refill_lanes idle lanes are given new (pixel, sample) items once this many are idle (or nothing else is runnable)
struct StageStats { uint32_t refill, trips, leaf, shade, refillLanes, shadeLanes; }; // wave-uniform, COUNT variant only

#define RTD_YIELD_DEFAULT 44
#define RTD_REFILL_DEFAULT 12

// Trace `total` items of the current unit.  Item i belongs to pixel slot map[i / per] (or i / per when map is null)
// and is sample s_base + i % per of that pixel; its colour is added to accumulator slot (sample < split ? 0 : 1).
//
// Every lane is a path slot in one of three states: IDLE (wants a new item), WALK (somewhere in the tree walk of its
// current ray; the walk state survives while the lane is parked), DONE (tree exhausted: wants the unbounded-object tests
// and Hittable.Reflection).  Per-ray work is heavy-tailed (tree nodes per ray: mean 26, p99 60, max 150), so running
// each stage until its slowest lane finishes leaves ~70 % of the lanes idle.  Instead a stage runs while enough lanes
// want it and yields to the stage that has collected the most waiting lanes; a ray with a long walk simply stays in WALK
// across several rounds.  Which lane computes what when has no effect on any result (streams are per item).
template <bool LDS, bool COUNT, bool COST>
RTD_INLINE void run_items(const RenderParams &p, const SceneView<LDS> &sc, RTD_AS3 uint32_t *acc, const RTD_AS3 uint32_t *pix,
                          const RTD_AS3 uint32_t *live, bool use_live, uint32_t total, uint32_t per, uint32_t s_base,
                          uint32_t split, Counters &cnt, StageStats &ss) {
    enum { IDLE = 0, WALK = 1, DONE = 2 };
    int st = IDLE;
    V3 o = mk(0, 0, 0), d = mk(0, 0, 0);
    const int end = sc.end;
    Walk w; walk_begin(w, sc.first); w.off = end; // idle lanes are parked at `end`
    Rng rng; rng.x = rng.y = rng.z = rng.w = 0;
    uint32_t colour = 0, slotOff = 0;
    int bounces = 0;
    uint32_t next = 0; // wave-uniform
    const bool fastDiv = total < (1u << 22) && per < (1u << 23); // see div_uniform
    const float perRcp = 1.0f / (float) per;
    for (;;) {
        // ---- refill: idle lanes take the next items of the unit (Scene.traceOnce's ray, Scene.fs:129-150) ----
        const unsigned long long idle = __builtin_amdgcn_ballot_w64(st == IDLE);
        const unsigned long long busy = ~idle;
        if (idle != 0ull && next < total && (__popcll(idle) >= p.refill_lanes || busy == 0ull)) {
            uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t) (idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) idle, 0u));
            uint32_t item = next + rank;
            if (COUNT) { ss.refill++; ss.refillLanes += (uint32_t) __popcll(idle); }
            if (st == IDLE && item < total) {
                uint32_t j = fastDiv ? div_uniform(item, per, perRcp) : item / per;
                uint32_t s = s_base + (item - j * per);
                uint32_t slot = use_live ? live[j] : j;
                int row = (int) pix[slot * 4 + 0], col = (int) pix[slot * 4 + 1];
                uint64_t pkey = (uint64_t) pix[slot * 4 + 2] | ((uint64_t) pix[slot * 4 + 3] << 32);
                rng = stream_for(pkey, s);
                slotOff = (((s < split) ? 0u : (uint32_t) p.chunk * 3u) + slot * 3u) | (slot << 16); // low half: acc word, high half: pixel slot
                colour = RTD_WHITE;
                bounces = 0;
                const CameraParams *cp = p.cam_ptr;
                asm volatile("" : "+s"(cp)); // keep the camera's 32 dwords out of the loop-carried SGPR set
                if (camera_ray(*cp, row, col, rng, o, d)) {
                    st = WALK;
                    walk_begin(w, sc.first);
                    if (COUNT) cnt.rays++;
                }
                // else: Scene.fs:144's ValueOption.get would throw; the sample counts as Black (adds nothing)
            }
            next += (uint32_t) __popcll(idle);
            next = __builtin_amdgcn_readfirstlane(next);
        }
        if (__builtin_amdgcn_ballot_w64(st != IDLE) == 0ull) {
            if (next >= total) break;
            continue;
        }

        // ---- walk: BoundingBox.hits over the tree image, leaf tests deferred out of the node loop ----
        // A lane steps iff w.off < end: finished walks sit at >= end, pending leaves carry RTD_LEAF (> end), idle lanes are
        // parked at `end`.  So one compare gives the active set, and waiting = busy - active.
        const int nBusy = __popcll(__builtin_amdgcn_ballot_w64(st != IDLE));
        if (__builtin_amdgcn_ballot_w64(st == WALK) != 0ull) {
            WalkCtx c = walk_ctx(d, w); // cheap to re-derive; keeps 9 doubles out of the parked state
            const int stop = (nBusy - p.yield_lanes) > 0 ? (nBusy - p.yield_lanes) : 0; // active <= stop  <=>  waiting >= yield
            for (;;) {
                for (;;) {
                    const bool act = w.off < end;
                    const int nAct = __popcll(__builtin_amdgcn_ballot_w64(act));
                    if (nAct <= stop) break;
                    if (COUNT) ss.trips++;
                    if (act) {
                        if (COUNT) cnt.aabb++;
                        node_step<LDS>(sc, o, c, w);
                    }
                }
                if (COUNT && __builtin_amdgcn_ballot_w64((w.off & RTD_LEAF) != 0) != 0ull) ss.leaf++;
                if (w.off & RTD_LEAF) {
                    if (COUNT) cnt.prim++;
                    leaf_test<LDS>(sc, o, d, c, w);
                }
                const bool fin = (st == WALK) && (w.off >= end);
                if (fin) st = DONE;
                const int nWalk = __popcll(__builtin_amdgcn_ballot_w64(st == WALK));
                const int nDone = __popcll(__builtin_amdgcn_ballot_w64(st == DONE));
                if (nWalk == 0 || nDone >= p.yield_lanes) break;
            }
        }

        // ---- finish: the rest of Scene.hitObject, then Hittable.Reflection (Scene.fs:77-112) ----
        if (COUNT) { const unsigned long long dm = __builtin_amdgcn_ballot_w64(st == DONE); if (dm) { ss.shade++; ss.shadeLanes += (uint32_t) __popcll(dm); } }
        if (st == DONE) {
            unbounded_tests<LDS, COUNT>(sc, o, d, w, cnt);
            bool done = false;
            uint32_t result = RTD_BLACK;
            if (w.best < 0) done = true; // "never heard from again": Black
            else {
                V3 strike = walk(o, d, w.bestLen); // Ray.walkAlong ray bestLength (Scene.fs:91)
                if (COUNT) cnt.refl++;
                if (reflection<LDS>(sc, w.best, strike, o, d, colour, rng)) { done = true; result = colour; }
                else {
                    bounces = bounces + 1;
                    if (bounces > p.depth) { done = true; result = RTD_HOTPINK; } // Scene.fs:98,114
                }
            }
            if (done) {
                const uint32_t aoff = slotOff & 0xFFFFu;
                if (result != 0u) { // PixelStats.add (Pixel.fs:97-101); Count is implied by the item count
                    lds_add(acc + aoff + 0, result & 0xFFu);
                    lds_add(acc + aoff + 1, (result >> 8) & 0xFFu);
                    lds_add(acc + aoff + 2, (result >> 16) & 0xFFu);
                }
                if (COST) lds_add(acc + 11u * (uint32_t) p.chunk + (slotOff >> 16), (uint32_t) bounces + 1u); // Scene.hitObject calls of this path
                st = IDLE;
                w.off = end;
            } else {
                st = WALK;
                walk_begin(w, sc.first);
                if (COUNT) cnt.rays++;
            }
        }
    }
}

// Pass B (MODE 2): phase 2 for the pixels of the cost-ordered list, STREAMED.  A wave reserves a run of list entries ("range"),
// hands out its npx*n2 items, and while the last paths of that range are still in flight it already reserves the next range and
// hands out its items: two accumulator slots alternate, a range is flushed (its sums added to what pass A left in `accum`) when
// its last path has ended.  There is no dependency between ranges, so no lane waits at a range boundary -- which is what makes
// small ranges (good load balance across waves) affordable.  Lane states and stage scheduling are those of run_items.
template <bool LDS, bool COUNT>
RTD_INLINE void run_stream(const RenderParams &p, const SceneView<LDS> &sc, RTD_AS3 uint32_t *wv, uint32_t n1, uint32_t n2, Counters &cnt,
                           StageStats &ss, uint64_t &sampleCount) {
    const int lane = threadIdx.x & 63;
    const uint32_t P = (uint32_t) p.chunk;
    const uint32_t SW = 7u * P; // words per slot: acc [P][3] then pix [P][4]
    const unsigned long long nList = (unsigned long long) *p.live_count;
    enum { IDLE = 0, WALK = 1, DONE = 2 };
    int st = IDLE;
    V3 o = mk(0, 0, 0), d = mk(0, 0, 0);
    const int end = sc.end;
    Walk w; walk_begin(w, sc.first); w.off = end;
    Rng rng; rng.x = rng.y = rng.z = rng.w = 0;
    uint32_t colour = 0, slotOff = 0; // word offset from wv of the path's accumulator triple (>= SW: slot 1)
    int bounces = 0;

    // wave-uniform: the range being handed out (cur) and the one draining (prev)
    unsigned long long curFirst = 0, prevFirst = 0;
    uint32_t curNpx = 0, curNext = 0, curTotal = 0, curOut = 0, curSlot = 1;
    uint32_t prevNpx = 0, prevOut = 0, prevSlot = 0;
    bool exhausted = false;
    const bool fastDiv = (uint64_t) P * (uint64_t) n2 < (1ull << 22) && n2 < (1u << 23); // a range holds at most P * n2 items; see div_uniform
    const float perRcp = 1.0f / (float) n2;

    auto flush = [&](unsigned long long first, uint32_t npx, uint32_t slot) {
        RTD_AS3 uint32_t *acc = wv + slot * SW;
        if ((uint32_t) lane < npx) { // this wave is the only writer of these pixels
            const unsigned long long lp = (unsigned long long) p.live_list[first + (uint32_t) lane];
            const i4 prev = ((const i4 *) p.accum)[lp];
            i4 out;
            out.x = prev.x + (int) n2;
            out.y = prev.y + (int) lds_take(acc + lane * 3 + 0);
            out.z = prev.z + (int) lds_take(acc + lane * 3 + 1);
            out.w = prev.w + (int) lds_take(acc + lane * 3 + 2);
            sampleCount += (uint64_t) n2;
            ((i4 *) p.accum)[lp] = out;
            if (p.rgb) {
                p.rgb[lp * 3 + 0] = (uint8_t) (out.y / out.x);
                p.rgb[lp * 3 + 1] = (uint8_t) (out.z / out.x);
                p.rgb[lp * 3 + 2] = (uint8_t) (out.w / out.x);
            }
        }
        __builtin_amdgcn_wave_barrier();
    };

    for (;;) {
        // ---- ranges: flush what has drained, reserve the next one when the current one has no items left ----
        if (prevNpx != 0u && prevOut == 0u) { flush(prevFirst, prevNpx, prevSlot); prevNpx = 0u; }
        if (curNext >= curTotal && prevNpx == 0u) {
            if (curNpx != 0u) { // the current range becomes the draining one (or is flushed at once if nothing is in flight)
                if (curOut == 0u) flush(curFirst, curNpx, curSlot);
                else { prevFirst = curFirst; prevNpx = curNpx; prevOut = curOut; prevSlot = curSlot; }
                curNpx = 0u; curNext = curTotal = curOut = 0u;
            }
            if (!exhausted) {
                unsigned long long first = 0;
                uint32_t npx = 0;
                if (lane == 0) { // guided self-scheduling over the cost-ordered list: big ranges first, single pixels at the end
                    const unsigned long long seen = __hip_atomic_load(p.queue_b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (seen < nList) {
                        unsigned long long want = (nList - seen) / (2ull * p.total_waves);
                        want = want < 1ull ? 1ull : (want > (unsigned long long) P ? (unsigned long long) P : want);
                        first = atomicAdd(p.queue_b, want);
                        if (first < nList) npx = (uint32_t) ((nList - first < want) ? (nList - first) : want);
                    }
                }
                npx = __builtin_amdgcn_readfirstlane(npx);
                if (npx == 0u) exhausted = true;
                else {
                    first = ((unsigned long long) __builtin_amdgcn_readfirstlane((uint32_t) (first >> 32)) << 32) |
                            __builtin_amdgcn_readfirstlane((uint32_t) first);
                    if (prevNpx != 0u) curSlot = prevSlot ^ 1u; // the draining range keeps its slot
                    RTD_AS3 uint32_t *acc = wv + curSlot * SW;
                    RTD_AS3 uint32_t *pix = acc + 3u * P;
                    for (uint32_t i = (uint32_t) lane; i < 3u * P; i += 64u) acc[i] = 0u;
                    if ((uint32_t) lane < npx) {
                        const unsigned long long lp = (unsigned long long) p.live_list[first + (uint32_t) lane];
                        uint32_t lr = (uint32_t) (lp / (unsigned long long) p.cols);
                        uint32_t c = (uint32_t) (lp - (unsigned long long) lr * (unsigned long long) p.cols);
                        uint32_t r = (uint32_t) p.row_first + lr * (uint32_t) p.row_stride;
                        uint64_t pkey = pixel_key(p.seed_key, (uint64_t) r * (uint64_t) p.cols + c); // global pixel index
                        pix[lane * 4 + 0] = (uint32_t) (p.max_h - (int) r - 1);
                        pix[lane * 4 + 1] = (uint32_t) ((int) c - p.max_w);
                        pix[lane * 4 + 2] = (uint32_t) pkey;
                        pix[lane * 4 + 3] = (uint32_t) (pkey >> 32);
                    }
                    __builtin_amdgcn_wave_barrier();
                    curFirst = first; curNpx = npx; curNext = 0u; curTotal = npx * n2; curOut = 0u;
                }
            }
        }
        const unsigned long long idle = __builtin_amdgcn_ballot_w64(st == IDLE);
        if (curNpx == 0u && prevNpx == 0u && exhausted && idle == ~0ull) break;

        // ---- refill from the current range ----
        {
            const uint32_t nIdle = (uint32_t) __popcll(idle);
            const uint32_t avail = curTotal - curNext;
            if (nIdle != 0u && avail != 0u && ((int) nIdle >= p.refill_lanes || nIdle == 64u)) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t) (idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) idle, 0u));
                if (COUNT) { ss.refill++; ss.refillLanes += nIdle; }
                const uint32_t take = nIdle < avail ? nIdle : avail;
                bool started = false;
                if (st == IDLE && rank < take) {
                    const uint32_t item = curNext + rank;
                    const uint32_t j = fastDiv ? div_uniform(item, n2, perRcp) : item / n2;
                    const uint32_t smp = n1 + (item - j * n2);
                    const RTD_AS3 uint32_t *pix = wv + curSlot * SW + 3u * P;
                    const int row = (int) pix[j * 4 + 0], col = (int) pix[j * 4 + 1];
                    const uint64_t pkey = (uint64_t) pix[j * 4 + 2] | ((uint64_t) pix[j * 4 + 3] << 32);
                    rng = stream_for(pkey, smp);
                    slotOff = curSlot * SW + j * 3u;
                    colour = RTD_WHITE;
                    bounces = 0;
                    const CameraParams *cp = p.cam_ptr;
                    asm volatile("" : "+s"(cp));
                    if (camera_ray(*cp, row, col, rng, o, d)) {
                        st = WALK;
                        walk_begin(w, sc.first);
                        started = true;
                        if (COUNT) cnt.rays++;
                    }
                }
                curNext += take;
                curOut += (uint32_t) __popcll(__builtin_amdgcn_ballot_w64(started));
            }
        }
        const int nBusy = __popcll(__builtin_amdgcn_ballot_w64(st != IDLE));
        if (nBusy == 0) continue;

        // ---- walk (as in run_items) ----
        if (__builtin_amdgcn_ballot_w64(st == WALK) != 0ull) {
            WalkCtx c = walk_ctx(d, w);
            const int stop = (nBusy - p.yield_lanes) > 0 ? (nBusy - p.yield_lanes) : 0;
            for (;;) {
                for (;;) {
                    const bool act = w.off < end;
                    const int nAct = __popcll(__builtin_amdgcn_ballot_w64(act));
                    if (nAct <= stop) break;
                    if (COUNT) ss.trips++;
                    if (act) {
                        if (COUNT) cnt.aabb++;
                        node_step<LDS>(sc, o, c, w);
                    }
                }
                if (COUNT && __builtin_amdgcn_ballot_w64((w.off & RTD_LEAF) != 0) != 0ull) ss.leaf++;
                if (w.off & RTD_LEAF) {
                    if (COUNT) cnt.prim++;
                    leaf_test<LDS>(sc, o, d, c, w);
                }
                const bool fin = (st == WALK) && (w.off >= end);
                if (fin) st = DONE;
                const int nWalk = __popcll(__builtin_amdgcn_ballot_w64(st == WALK));
                const int nDone = __popcll(__builtin_amdgcn_ballot_w64(st == DONE));
                if (nWalk == 0 || nDone >= p.yield_lanes) break;
            }
        }

        // ---- finish (as in run_items) ----
        if (COUNT) { const unsigned long long dm = __builtin_amdgcn_ballot_w64(st == DONE); if (dm) { ss.shade++; ss.shadeLanes += (uint32_t) __popcll(dm); } }
        bool ended = false;
        if (st == DONE) {
            unbounded_tests<LDS, COUNT>(sc, o, d, w, cnt);
            uint32_t result = RTD_BLACK;
            if (w.best < 0) ended = true;
            else {
                V3 strike = walk(o, d, w.bestLen);
                if (COUNT) cnt.refl++;
                if (reflection<LDS>(sc, w.best, strike, o, d, colour, rng)) { ended = true; result = colour; }
                else {
                    bounces = bounces + 1;
                    if (bounces > p.depth) { ended = true; result = RTD_HOTPINK; }
                }
            }
            if (ended) {
                if (result != 0u) {
                    lds_add(wv + slotOff + 0, result & 0xFFu);
                    lds_add(wv + slotOff + 1, (result >> 8) & 0xFFu);
                    lds_add(wv + slotOff + 2, (result >> 16) & 0xFFu);
                }
                st = IDLE;
                w.off = end;
            } else {
                st = WALK;
                walk_begin(w, sc.first);
                if (COUNT) cnt.rays++;
            }
        }
        const bool inCur = (curNpx != 0u) && ((slotOff >= SW) == (curSlot == 1u));
        curOut -= (uint32_t) __popcll(__builtin_amdgcn_ballot_w64(ended && inCur));
        prevOut -= (uint32_t) __popcll(__builtin_amdgcn_ballot_w64(ended && !inCur));
    }
}

// MODE 0: fused -- a unit's pixels go through phase 1, the adaptive decision and phase 2 on one wave.
// MODE 1: pass A -- phase 1 and the decision for every pixel; pixels that continue are appended to `pairs` with the number of
//         rays their 2k+1 samples took (a cost estimate), the others are final.
// MODE 2: pass B -- phase 2 for the pixels of `live_list`, which the host-side launch sequence has ordered by decreasing cost
//         (longest job first); unit size shrinks with the remaining list, down to one pixel.
// Per-pixel cost is heavy-tailed (a pixel on a glass sphere: ~20 rays per sample, 4 ms of one wave), so when a shard has only a few
// units per wave the fused kernel ends with most waves waiting for a few long units started late; A + sort + B removes that tail.
// Every mode computes the same integers: which wave traces which sample when has no effect (streams are per item).
template <bool LDS, bool COUNT, int BLOCK, int MODE>
__global__ void __launch_bounds__(BLOCK) render_kernel(const RenderParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

    uint32_t sceneBytes = 0;
    if (LDS) { // stage the scene image: 16 B per lane per trip, coalesced
        sceneBytes = p.off.total;
        const d2 *src = (const d2 *) p.scene_image;
        RTD_AS3 d2 *dst = (RTD_AS3 d2 *) smem;
        for (uint32_t i = threadIdx.x; i < sceneBytes / 16u; i += BLOCK) dst[i] = src[i];
        __syncthreads();
    }
    const SceneView<LDS> sc = make_view<LDS>(p, smem);
    if (LDS) { // make the links absolute LDS addresses: a walk position then IS the record's address (no add per visit)
        RTD_AS3 unsigned char *nodes = (RTD_AS3 unsigned char *) smem + p.off.node;
        for (int i = threadIdx.x; i < p.off.n_nodes; i += BLOCK) {
            RTD_AS3 i2 *lk = (RTD_AS3 i2 *) (nodes + i * RTD_NODE_BYTES + 72);
            i2 v = *lk;
            if (!(v.x & RTD_LEAF)) v.x += sc.first;
            v.y += sc.first;
            *lk = v;
        }
        __syncthreads();
    }
    const uint32_t P = (uint32_t) p.chunk;
    RTD_AS3 uint32_t *wv = (RTD_AS3 uint32_t *) (smem + sceneBytes) + (size_t) wave * RTD_WAVE_WORDS(P);
    RTD_AS3 uint32_t *acc = wv;
    RTD_AS3 uint32_t *pix = wv + 6u * P;
    RTD_AS3 uint32_t *live = pix + 4u * P;

    const uint64_t nLocal = (uint64_t) p.n_rows * (uint64_t) p.cols;
    const uint32_t k = (uint32_t) p.k;
    const uint32_t n1 = 2u * k + 1u;
    const int n2s = p.spp - 2 * p.k - 1; // Scene.fs:191
    const uint32_t n2 = n2s > 0 ? (uint32_t) n2s : 0u;

    const unsigned long long tStart = COUNT ? __builtin_amdgcn_s_memrealtime() : 0ull;
    Counters cnt; cnt.rays = cnt.aabb = cnt.prim = cnt.refl = 0;
    StageStats ss; ss.refill = ss.trips = ss.leaf = ss.shade = ss.refillLanes = ss.shadeLanes = 0;
    uint32_t earlyCount = 0;
    uint64_t sampleCount = 0; // Scene.traceOnce calls = sum of PixelStats.Count

    if (MODE == 2) run_stream<LDS, COUNT>(p, sc, wv, n1, n2, cnt, ss, sampleCount);
    else
    for (;;) {
        unsigned long long first = 0;
        uint32_t npx = 0;
        if (MODE == 2) {
            const unsigned long long nList = 0ull; // (pass B runs run_stream; this branch is dead code kept for the template)
            if (lane == 0) {
                const unsigned long long seen = __hip_atomic_load(p.queue_b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (seen < nList) {
                    unsigned long long want = (nList - seen) / (2ull * p.total_waves);
                    want = want < 1ull ? 1ull : (want > (unsigned long long) P ? (unsigned long long) P : want);
                    first = atomicAdd(p.queue_b, want);
                    if (first < nList) npx = (uint32_t) ((nList - first < want) ? (nList - first) : want);
                }
            }
            npx = __builtin_amdgcn_readfirstlane(npx);
            first = ((unsigned long long) __builtin_amdgcn_readfirstlane((uint32_t) (first >> 32)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t) first);
        } else {
            uint32_t unit = 0;
            if (lane == 0) unit = atomicAdd(p.queue, 1u);
            unit = __builtin_amdgcn_readfirstlane(unit);
            first = (unsigned long long) unit * P;
            if (first < nLocal) npx = (uint32_t) ((nLocal - first < (unsigned long long) P) ? (nLocal - first) : (unsigned long long) P);
        }
        if (npx == 0u) break;

        // per-pixel coordinates (Scene.fs:219,226) and stream key; clear the accumulators
        for (uint32_t i = (uint32_t) lane; i < 6u * P; i += 64u) acc[i] = 0u;
        if (MODE == 1 && (uint32_t) lane < P) acc[11u * P + lane] = 0u;
        unsigned long long lp = 0; // local pixel of lane j < npx
        if ((uint32_t) lane < npx) {
            lp = (MODE == 2) ? (unsigned long long) p.live_list[first + (uint32_t) lane] : first + (uint32_t) lane;
            uint32_t lr = (uint32_t) (lp / (unsigned long long) p.cols);
            uint32_t c = (uint32_t) (lp - (unsigned long long) lr * (unsigned long long) p.cols);
            uint32_t r = (uint32_t) p.row_first + lr * (uint32_t) p.row_stride;
            uint64_t pkey = pixel_key(p.seed_key, (uint64_t) r * (uint64_t) p.cols + c); // global pixel index
            pix[lane * 4 + 0] = (uint32_t) (p.max_h - (int) r - 1);
            pix[lane * 4 + 1] = (uint32_t) ((int) c - p.max_w);
            pix[lane * 4 + 2] = (uint32_t) pkey;
            pix[lane * 4 + 3] = (uint32_t) (pkey >> 32);
        }
        __builtin_amdgcn_wave_barrier();

        int sumR = 0, sumG = 0, sumB = 0, count = 0;
        bool cont = false;
        uint32_t nLive = 0;
        if (MODE != 2) {
            // ---- phase 1: 2k+1 samples per pixel, sums split after sample k (Scene.fs:172-182) ----
            run_items<LDS, COUNT, MODE == 1>(p, sc, acc, pix, live, false, npx * n1, n1, 0u, k + 1u, cnt, ss);
            __builtin_amdgcn_wave_barrier();

            // ---- decide (Scene.fs:177-188) and compact the pixels that continue ----
            if ((uint32_t) lane < npx) {
                int aR = (int) lds_take(acc + lane * 3 + 0), aG = (int) lds_take(acc + lane * 3 + 1), aB = (int) lds_take(acc + lane * 3 + 2);
                int bR = (int) lds_take(acc + P * 3 + lane * 3 + 0), bG = (int) lds_take(acc + P * 3 + lane * 3 + 1),
                    bB = (int) lds_take(acc + P * 3 + lane * 3 + 2);
                int c1 = (int) k + 1;
                count = (int) n1;
                sumR = aR + bR; sumG = aG + bG; sumB = aB + bB;
                // PixelStats.mean (Pixel.fs:103-108) is integer division; Pixel.difference (Pixel.fs:113-116) is L1
                int oR = (aR / c1) & 0xFF, oG = (aG / c1) & 0xFF, oB = (aB / c1) & 0xFF;
                int nR = (sumR / count) & 0xFF, nG = (sumG / count) & 0xFF, nB = (sumB / count) & 0xFF;
                int diff = abs(nR - oR) + abs(nG - oG) + abs(nB - oB);
                if (diff == 0) earlyCount++;
                cont = (diff != 0) && (n2 > 0u);
            }
            const unsigned long long liveMask = __builtin_amdgcn_ballot_w64(cont);
            nLive = (uint32_t) __popcll(liveMask);
            const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t) (liveMask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) liveMask, 0u));
            if (MODE == 0) {
                if (cont) live[pos] = (uint32_t) lane;
            } else if (nLive > 0u) { // pass A: hand the pixel over to pass B, with its cost estimate
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(p.live_count, nLive);
                base = __builtin_amdgcn_readfirstlane(base);
                if (cont) p.pairs[base + pos] = ((unsigned long long) acc[11u * P + lane] << 32) | (unsigned long long) (uint32_t) lp;
            }
            __builtin_amdgcn_wave_barrier();
        }

        // ---- phase 2: the remaining spp-2k-1 samples of the surviving pixels (Scene.fs:191-192) ----
        if (MODE == 0 && nLive > 0u) {
            run_items<LDS, COUNT, false>(p, sc, acc, pix, live, true, nLive * n2, n2, n1, 0xFFFFFFFFu, cnt, ss);
            __builtin_amdgcn_wave_barrier();
        }
        if (MODE == 2) {
            run_items<LDS, COUNT, false>(p, sc, acc, pix, live, false, npx * n2, n2, n1, 0xFFFFFFFFu, cnt, ss);
            __builtin_amdgcn_wave_barrier();
        }

        // ---- PixelStats and mean out: one 16-byte store per pixel ----
        if ((uint32_t) lane < npx) {
            if (MODE == 2) { // add phase 2 to what pass A left (this wave is the only writer of the pixel)
                const i4 prev = ((const i4 *) p.accum)[lp];
                sumR = prev.y + (int) lds_take(acc + lane * 3 + 0);
                sumG = prev.z + (int) lds_take(acc + lane * 3 + 1);
                sumB = prev.w + (int) lds_take(acc + lane * 3 + 2);
                count = prev.x + (int) n2;
                sampleCount += (uint64_t) n2;
            } else {
                if (MODE == 0 && cont) {
                    sumR += (int) lds_take(acc + lane * 3 + 0);
                    sumG += (int) lds_take(acc + lane * 3 + 1);
                    sumB += (int) lds_take(acc + lane * 3 + 2);
                    count += (int) n2;
                }
                sampleCount += (uint64_t) count;
            }
            i4 out; out.x = count; out.y = sumR; out.z = sumG; out.w = sumB;
            ((i4 *) p.accum)[lp] = out;
            if (p.rgb) {
                p.rgb[lp * 3 + 0] = (uint8_t) (sumR / count);
                p.rgb[lp * 3 + 1] = (uint8_t) (sumG / count);
                p.rgb[lp * 3 + 2] = (uint8_t) (sumB / count);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }

    // ---- counters: one wave reduction and <= 6 atomics per wave per launch ----
    uint64_t e = wave_sum_u64(earlyCount), s = wave_sum_u64(sampleCount);
    if (COUNT) {
        uint64_t a = wave_sum_u64(cnt.rays), b = wave_sum_u64(cnt.aabb), c = wave_sum_u64(cnt.prim), dd = wave_sum_u64(cnt.refl);
        if (lane == 0) {
            atomicAdd(&p.counters[0], (unsigned long long) a);
            atomicAdd(&p.counters[1], (unsigned long long) b);
            atomicAdd(&p.counters[2], (unsigned long long) c);
            atomicAdd(&p.counters[3], (unsigned long long) dd);
            atomicAdd(&p.counters[8], (unsigned long long) ss.refill);
            atomicAdd(&p.counters[9], (unsigned long long) ss.trips);
            atomicAdd(&p.counters[10], (unsigned long long) ss.leaf);
            atomicAdd(&p.counters[11], (unsigned long long) ss.shade);
            atomicAdd(&p.counters[12], (unsigned long long) ss.refillLanes);
            atomicAdd(&p.counters[13], (unsigned long long) ss.shadeLanes);
            const unsigned long long tEnd = __builtin_amdgcn_s_memrealtime();
            atomicAdd(&p.counters[14], tEnd - tStart);                         // sum of wave lifetimes
            atomicMax(&p.counters[15], tEnd);                                  // last wave to finish
            atomicMax(&p.counters[7], 0x4000000000000000ull - tStart);         // (2^62 - earliest start)
        }
    }
    if (lane == 0) {
        atomicAdd(&p.counters[4], (unsigned long long) s);
        atomicAdd(&p.counters[5], (unsigned long long) e);
    }
}


// ---- ordering of the pass-B list: bucket sort of (cost, pixel) pairs, heaviest first -------------------------------------------
// 64 buckets over rays per phase-1 sample (x4); order inside a bucket is arbitrary (it only changes which wave traces what).
#define RTD_COST_BUCKETS 64
RTD_INLINE uint32_t cost_bucket(unsigned long long pair, uint32_t n1) {
    const uint32_t cost = (uint32_t) (pair >> 32);
    const uint32_t b = (cost * 4u) / n1;
    return b >= RTD_COST_BUCKETS ? RTD_COST_BUCKETS - 1u : b;
}
// Both kernels: one workgroup owns a contiguous slice of the list and counts it in LDS first, so the global atomics are one
// per (workgroup, bucket) -- per-item global atomics on 64 addresses serialise (measured: 0.78 ms for 174 k pairs when most
// pixels share a bucket).  The order inside a bucket is arbitrary; no result depends on it.
RTD_INLINE void sort_slice(unsigned int n, unsigned int &lo, unsigned int &hi) {
    const unsigned int per = (n + gridDim.x - 1u) / gridDim.x;
    lo = blockIdx.x * per;
    hi = lo + per < n ? lo + per : n;
    if (lo > n) lo = n;
}
__global__ void sort_hist_kernel(const unsigned long long *pairs, const unsigned int *count, uint32_t n1, unsigned int *hist) {
    __shared__ unsigned int local[RTD_COST_BUCKETS];
    if (threadIdx.x < RTD_COST_BUCKETS) local[threadIdx.x] = 0u;
    __syncthreads();
    unsigned int lo, hi;
    sort_slice(*count, lo, hi);
    for (unsigned int i = lo + threadIdx.x; i < hi; i += blockDim.x) atomicAdd(&local[cost_bucket(pairs[i], n1)], 1u);
    __syncthreads();
    if (threadIdx.x < RTD_COST_BUCKETS && local[threadIdx.x] != 0u) atomicAdd(&hist[threadIdx.x], local[threadIdx.x]);
}
__global__ void sort_offsets_kernel(const unsigned int *hist, unsigned int *offsets) { // descending cost: bucket 63 first
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        unsigned int run = 0;
        for (int b = RTD_COST_BUCKETS - 1; b >= 0; --b) { offsets[b] = run; run += hist[b]; }
    }
}
__global__ void sort_scatter_kernel(const unsigned long long *pairs, const unsigned int *count, uint32_t n1, const unsigned int *offsets,
                                    unsigned int *cursor, unsigned int *list) {
    __shared__ unsigned int local[RTD_COST_BUCKETS], base[RTD_COST_BUCKETS];
    if (threadIdx.x < RTD_COST_BUCKETS) local[threadIdx.x] = 0u;
    __syncthreads();
    unsigned int lo, hi;
    sort_slice(*count, lo, hi);
    for (unsigned int i = lo + threadIdx.x; i < hi; i += blockDim.x) atomicAdd(&local[cost_bucket(pairs[i], n1)], 1u);
    __syncthreads();
    if (threadIdx.x < RTD_COST_BUCKETS) { // reserve this workgroup's range of every bucket it holds items of
        const unsigned int c = local[threadIdx.x];
        base[threadIdx.x] = offsets[threadIdx.x] + (c != 0u ? atomicAdd(&cursor[threadIdx.x], c) : 0u);
        local[threadIdx.x] = 0u;
    }
    __syncthreads();
    for (unsigned int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const unsigned long long pr = pairs[i];
        const uint32_t b = cost_bucket(pr, n1);
        list[base[b] + atomicAdd(&local[b], 1u)] = (unsigned int) (pr & 0xFFFFFFFFull);
    }
}

} // namespace rtd
