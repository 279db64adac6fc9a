// rt_render_kernel.h -- the persistent render megakernel for gfx950.
//
// Takes the role of Scene.render / renderPixel / traceOnce / traceRay (Scene.fs:93-236) for one shard of image rows.
//
// Execution model:
//   * One workgroup per CU-slot, the tree, object geometry and meta words staged ONCE per workgroup into LDS (<= ~146 KiB at 1024
//     threads and 16-pixel units; the material table stays in global memory), then every WAVE is an independent worker pulling
//     work units (runs of pixels) from a global queue, one atomic per unit.
//   * Inside a unit the 64 lanes are path slots in one of four states (idle / walking the tree / walk finished / waiting for
//     the general reflection); stages (refill, general reflection, node loop, leaf tests, shade) run when enough lanes want
//     them -- see Sched.
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
#define RTD_MAX_PARK 256
#define RTD_PARK_DEFAULT 96 /* entries of a wave's general pool (88 B each, global memory).  Bench frame, ms / GB written to HBM: 45: 124.6, 64: 109.7 / 14.5, 80: 108.8 / 24.0, 96: 108.2 / 28.1, 128: 108.2 / 30.7 (scripts/pool_traffic.sh) */
#ifndef RTD_PARK_L_DEFAULT
#define RTD_HYBRID_LANES 16 /* node_loop_glb32: this many lanes at LDS-held records make a trip of their own (measured, rt_device.h) */
#define RTD_PARK_L_DEFAULT 64 /* entries of a wave's pool of parked Lambert hits; 0: Lambert hits are shaded where they fall */
#endif

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
    int32_t leaf_wait;             // the node loop hands over to a leaf pass once this many lanes wait (>= yield_lanes, or a walk stage could spin)
    int32_t yield_lanes;           // see Sched: a stage yields once this many lanes wait for another stage
    int32_t refill_lanes;          // see Sched: idle lanes are refilled once this many are idle
    int32_t park;                  // entries of a wave's park pool (0: rare styles are shaded in place)
    int32_t park_l;                // entries of a wave's pool of parked LAMBERT hits (0: they are shaded where they fall)
    int32_t park_l_lds;            // 1: that pool lives in the workgroup's LDS (56-byte entries behind the waves' scratch), 0: in park_pool
    int32_t lds_node_bytes;        // LDS=false timed kernels ("hybrid"): bytes of the node32 section's first part held at the START of the LDS
    int32_t lds_node_thr;          // ... and how many lanes at those records make a trip of their own (node_loop_glb32; 1..65)
    int32_t *accum;                // [n_rows*cols][4]
    uint8_t *rgb;                  // [n_rows*cols][3] or null
    unsigned long long *counters;  // [16]: rays, aabb, prim, refl, samples, pixels_early, -, -, then stage executions (COUNT):
                                   //       [8] refill [9] node trips [10] leaf stages [11] shade stages [12] lanes refilled [13] lanes shaded
                                   //       and, 160 bytes in, [0] slow stages [1] lanes in them [2] lanes parked
    unsigned int *queue;           // work-unit counter of the fused kernel / of pass A, zeroed before launch
    unsigned char *park_pool;      // [waves of the grid][RTD_PARK_ENTRY_BYTES * (park + (park_l_lds ? 0 : park_l))]: per-wave pools of parked paths (see Sched)
    // two-pass rendering (see render_kernel): pass A appends (cost << 32 | local pixel) for every pixel that continues;
    // pass B walks `live_list` (those pixels ordered by decreasing cost) with its own queue
    unsigned long long *pairs;
    const unsigned int *live_list;
    unsigned int *live_count;      // number of pairs / list entries (device side)
    unsigned long long *queue_b;   // next unassigned list entry
    uint32_t total_waves;          // waves of the grid (guided unit sizes in pass B)
    // rt_scene_tune's probe: every ray whose stream-state hash has `ray_log_mask` clear is appended
    double *ray_log;               // [ray_log_cap][6]: origin, direction; null outside a probe
    unsigned int *ray_log_count;   // rays that wanted a slot (may exceed the capacity: the probe is then repeated more thinly)
    uint32_t ray_log_cap, ray_log_mask;
};

// Per-wave LDS scratch (in 4-byte words), P = pixels per work unit:
//   acc  [2][P][3]  sums of slot 0 / slot 1
//   pix  [P][4]     row, col, pixel stream key lo, hi
//   live [P]        compacted pixel slots for phase 2 (fused mode), or
//   cost [P]        rays traced for the pixel in phase 1 (pass A: the cost estimate that orders pass B; pass A has no phase 2)
//   cand [P][2]     the leaves the pixel's camera rays can reach, as two queue words (pixel_candidates, rt_device.h), or RTD_CAND_WALK
#define RTD_WAVE_WORDS(P) (18u * (uint32_t) (P)) /* fused: 13 P used; pass B: two slots of {acc [P][3], pix [P][4]}, then cand [2][P][2] */
#define RTD_WAVE_WORDS_A(P) (13u * (uint32_t) (P)) /* pass A: acc, pix, cost, cand [P][2] -- a tighter footprint, so its units can be wider */

// Wave-private LDS words: adds from many lanes may land on one word (same pixel), so they are ds_add_u32; the owner
// lane later takes the sum and clears the word in one ds_wrxchg.  One wave's LDS operations execute in order.
RTD_INLINE void lds_add(RTD_AS3 uint32_t *p, uint32_t v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
RTD_INLINE uint32_t lds_take(RTD_AS3 uint32_t *p) { return __hip_atomic_exchange(p, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }

RTD_INLINE uint64_t wave_sum_u64(uint64_t v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor((unsigned long long) v, off, 64);
    return v;
}
RTD_INLINE uint32_t lane_rank(unsigned long long mask) { // number of set bits of `mask` below this lane
    return __builtin_amdgcn_mbcnt_hi((uint32_t) (mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) mask, 0u));
}

// F32: the LDS copy is the timed variant's -- geo | meta | node32 (stage_scene<.., true>), walked by node_loop_lds32
template <bool LDS, bool F32 = false> RTD_INLINE SceneView<LDS> make_view(const RenderParams &p, const unsigned char *lds_base);
template <> RTD_INLINE SceneView<true> make_view<true, false>(const RenderParams &p, const unsigned char *lds_base) {
    SceneView<true> v;
    const RTD_AS3 unsigned char *b = (const RTD_AS3 unsigned char *) lds_base;
    v.node = (Ptrs<true>::bp) (b + p.off.node);
    v.geo = (Ptrs<true>::d2p) (b + p.off.geo);
    v.meta = (Ptrs<true>::i2p) (b + p.off.meta);
    v.mat = (const double *) (p.scene_image + p.off.mat);
    v.n_nodes = p.off.n_nodes; v.n_bounded = p.off.n_bounded; v.n_unbounded = p.off.n_unbounded;
    v.first = (int) (uint32_t) (uintptr_t) v.node; v.end = v.first + v.n_nodes * RTD_NODE_BYTES; // links were made absolute at staging
    v.tex = p.tex; v.texels = p.texels; v.lds_lim = 0; v.lds_thr = 65; v.narrow = (v.n_bounded + v.n_unbounded) < 16384 ? 1 : 0;
    return v;
}
template <> RTD_INLINE SceneView<true> make_view<true, true>(const RenderParams &p, const unsigned char *lds_base) {
    SceneView<true> v;
    const RTD_AS3 unsigned char *b = (const RTD_AS3 unsigned char *) lds_base - p.off.geo; // the copy starts at the image's `geo` section
    v.node = (Ptrs<true>::bp) (b + p.off.node32);
    v.geo = (Ptrs<true>::d2p) (b + p.off.geo);
    v.meta = (Ptrs<true>::i2p) (b + p.off.meta);
    v.mat = (const double *) (p.scene_image + p.off.mat);
    v.n_nodes = p.off.n_nodes; v.n_bounded = p.off.n_bounded; v.n_unbounded = p.off.n_unbounded;
    v.first = (int) (uint32_t) (uintptr_t) v.node; v.end = v.first + v.n_nodes * RTD_NODE32_BYTES;
    v.tex = p.tex; v.texels = p.texels; v.lds_lim = 0; v.lds_thr = 65; v.narrow = (v.n_bounded + v.n_unbounded) < 16384 ? 1 : 0;
    return v;
}
template <> RTD_INLINE SceneView<false> make_view<false, true>(const RenderParams &p, const unsigned char *) { // the timed variant over global memory
    SceneView<false> v;
    const unsigned char *b = p.scene_image;
    v.node = b + p.off.node32; // links are byte offsets from here (node_loop_glb32)
    v.geo = (const d2 *) (b + p.off.geo);
    v.meta = (const i2 *) (b + p.off.meta);
    v.mat = (const double *) (b + p.off.mat);
    v.n_nodes = p.off.n_nodes; v.n_bounded = p.off.n_bounded; v.n_unbounded = p.off.n_unbounded;
    v.first = 0; v.end = v.n_nodes * RTD_NODE32_BYTES;
    v.tex = p.tex; v.texels = p.texels; v.lds_lim = 0; v.lds_thr = 65; v.narrow = (v.n_bounded + v.n_unbounded) < 16384 ? 1 : 0;
    return v;
}
template <> RTD_INLINE SceneView<false> make_view<false, false>(const RenderParams &p, const unsigned char *) {
    SceneView<false> v;
    const unsigned char *b = p.scene_image;
    v.node = b + p.off.node;
    v.geo = (const d2 *) (b + p.off.geo);
    v.meta = (const i2 *) (b + p.off.meta);
    v.mat = (const double *) (b + p.off.mat);
    v.n_nodes = p.off.n_nodes; v.n_bounded = p.off.n_bounded; v.n_unbounded = p.off.n_unbounded;
    v.first = 0; v.end = v.n_nodes * RTD_NODE_BYTES;
    v.tex = p.tex; v.texels = p.texels; v.lds_lim = 0; v.lds_thr = 65; v.narrow = (v.n_bounded + v.n_unbounded) < 16384 ? 1 : 0;
    return v;
}

struct StageStats { // wave-uniform, COUNT variant only
    uint32_t refill, trips, leaf, shade, refillLanes, shadeLanes, slow, slowLanes, parkedLanes;
    unsigned long long tRefill, tSlow, tWalk, tShade; // shader-clock cycles this wave spent inside each stage (s_memtime)
    unsigned long long loopTrips, loopLanes;            // -DRTD_STAGE_CLOCKS only: trips of the timed variant's node loop, lanes that stepped in them
    unsigned long long tLoop, tLeaf, tUnb, tCam, tLamb; // -DRTD_STAGE_CLOCKS only: node loop / leaf passes (inside tWalk), unbounded tests (inside tShade),
                                                        // new items (inside tRefill), Lambert batches (inside tSlow)
};

// Diagnostic builds (-DRTD_STAGE_CLOCKS): the per-stage cycle sums of the counting variant in the timed variant too
#ifdef RTD_STAGE_CLOCKS
#define RTD_CLK true
#else
#define RTD_CLK false
#endif
#define RTD_YIELD_DEFAULT 50
#define RTD_LEAF_WAIT_EXTRA 5 /* RenderParams::leaf_wait = yield_lanes + this (at most 64) */
#define RTD_REFILL_DEFAULT 8

// ---- the lane scheduler shared by every render mode ------------------------------------------------------------------------
// Every lane of a wave is a path slot in one of four states: IDLE (wants a new item), WALK (somewhere in the tree walk of its
// current ray; the walk state survives while the lane is parked), DONE (tree exhausted: wants the unbounded-object tests and
// Hittable.Reflection), SLOW (its hit needs the general `reflection`: any style but an untextured light source or Lambert sphere).
// Per-ray work is heavy-tailed (tree nodes per ray: mean 24, p99 60, max 150), so running each stage until its slowest lane
// finishes leaves ~70 % of the lanes idle.  Instead a stage runs while enough lanes want it:
//   yield_lanes  the walk stage yields, and the shade stage runs, once this many walks are finished (or none is left walking)
//   leaf_wait    (yield_lanes + RTD_LEAF_WAIT_EXTRA) inside the walk stage the node loop hands over to a leaf pass once this many
//                lanes are waiting -- for a pending leaf test, or finished.  Never below yield_lanes: the stage would spin
//   refill_lanes idle lanes are given new work once this many are idle (or nothing else is runnable)
// The shade stage shades the two common cases where they fall (reflection_fast) -- about 250 instructions.  The other styles'
// code is twice as long and used to run in almost every shade stage for the two or three lanes that needed it; now such a
// path is PARKED: its state (88 bytes) goes to the wave's own pool in global memory (L2-resident: a few KB per wave in use)
// and the lane is free for other work.  When a refill finds at least as many parked paths as idle lanes (or no new items),
// the idle lanes take parked paths instead of new items and run the general `reflection` together.  A pool that is full
// (or switched off, p.park = 0) leaves the path in its lane and the general code runs for it at the next turn of the loop.
// Which lane computes what when has no effect on any result: streams are per item.
#define RTD_PARK_ENTRY_BYTES 96 /* 5 x 16 B + 8 B, padded */
#define RTD_PARK_L_LDS_BYTES 56 /* a parked Lambert hit in LDS: strike 24, rng 16, colour, slot, bounces | inside << 31, object */
enum { L_IDLE = 0, L_WALK = 1, L_DONE = 2, L_SLOW = 3, L_LAMB = 4, L_TEX = 5 };

// LOG: the kernel may be asked to log rays (rt_scene_tune's probe): only the fused mode's instantiations carry that code -- the
// log's four kernel arguments otherwise sit in scalar registers across the whole loop of the two-pass kernels, which have none to spare
template <bool LDS, bool COUNT, bool TEX, bool LOG>
struct Sched {
    const RenderParams &p;
    const SceneView<LDS> &sc;
    Counters &cnt;
    StageStats &ss;
    unsigned char *pool; // this wave's park pools in global memory: the general one, then (unless it is in LDS) the Lambert one
    RTD_AS3 unsigned char *poolLds; // this wave's Lambert pool in LDS (p.park_l_lds)
    uint32_t parked;     // entries in the general pool (wave-uniform)
    uint32_t parkedL;    // entries in the Lambert pool (wave-uniform)
    uint32_t parkedT;    // entries in the pool of hits whose colour is a parameterised texture (TEX kernels only; wave-uniform)
    const int end;
    // lane state
    int st;
    V3 o, d;
    Walk w;
    Rng rng;
    uint32_t colour, slotOff;
    int bounces;
    uint32_t pend; // queue of pending leaf tests (node_loop_lds32: two 16-bit entries; node_loop_glb32: the older full-width entry); always 0
                   // outside a walk and in the counting variant, which does not queue
    uint32_t texc; // L_SLOW lanes: the hit's texture colour if stage_tex has evaluated one, else RTD_NO_TEX
    // set by the stages for the caller's bookkeeping: this lane's path ended during the current turn, with this colour
    bool ended;
    uint32_t pend1; // node_loop_glb32's newer entry (declared apart from `pend`: as neighbours the two are stored together by one vector
                    // store, and the whole scheduler state then lives in scratch memory)
    uint32_t result;

    RTD_INLINE Sched(const RenderParams &p_, const SceneView<LDS> &sc_, Counters &cnt_, StageStats &ss_, unsigned char *pool_, RTD_AS3 unsigned char *poolLds_)
        : p(p_), sc(sc_), cnt(cnt_), ss(ss_), pool(pool_), poolLds(poolLds_), parked(0u), parkedL(0u), parkedT(0u), end(sc_.end) {
        st = L_IDLE;
        o = mk(0, 0, 0); d = mk(0, 0, 0);
        walk_begin(w, sc.first); w.off = end; // idle lanes are parked at `end`
        rng.x = rng.y = rng.z = rng.w = 0;
        colour = 0; slotOff = 0; bounces = 0; pend = 0u; pend1 = 0u; texc = RTD_NO_TEX;
        ended = false; result = 0;
    }

    // the probe of rt_scene_tune: a thinned-out log of the rays as they start (which rays: a hash of the ray's stream state, so
    // the logged SET does not depend on scheduling; the host sorts it)
    RTD_INLINE void log_ray() {
        if (!LOG) return;
        if (p.ray_log == nullptr) return; // wave-uniform: one scalar compare per ray outside a probe
        if (((((rng.x ^ rng.w) * 0x9E3779B1u) >> 8) & p.ray_log_mask) == 0u) {
            const unsigned int at = atomicAdd(p.ray_log_count, 1u);
            if (at < p.ray_log_cap) {
                double *r = p.ray_log + 6u * (size_t) at;
                r[0] = o.x; r[1] = o.y; r[2] = o.z; r[3] = d.x; r[4] = d.y; r[5] = d.z;
            }
        }
    }

    // a new (pixel, sample) item: Scene.traceOnce's ray (Scene.fs:129-150)
    RTD_INLINE bool start_item(uint64_t pkey, uint32_t sample, int row, int col, uint32_t slot_off, uint32_t cand, uint32_t cand2) {
        const unsigned long long k0 = RTD_CLK ? __builtin_amdgcn_s_memtime() : 0ull;
        struct Stamp { StageStats &ss; unsigned long long k0; RTD_INLINE ~Stamp() { if (RTD_CLK) ss.tCam += __builtin_amdgcn_s_memtime() - k0; } } stamp{ss, k0};
        rng = stream_for(pkey, sample);
        slotOff = slot_off;
        colour = RTD_WHITE;
        bounces = 0;
        const CameraParams *cp = p.cam_ptr;
        asm volatile("" : "+s"(cp)); // keep the camera's 32 dwords out of the loop-carried SGPR set
        if (camera_ray(*cp, row, col, rng, o, d)) {
            st = L_WALK;
            walk_begin(w, sc.first);
            if (!COUNT && cand != RTD_CAND_WALK) { // the tree walk of this pixel's camera rays was made once (pixel_candidates)
                w.off = end; pend = cand; pend1 = cand2;
                if (cand == 0u) st = L_DONE; // nothing in reach: straight to the unbounded objects
            }
            if (COUNT) cnt.rays++;
            log_ray();
            return true;
        }
        return false; // Scene.fs:144's ValueOption.get would throw; the sample counts as Black (adds nothing)
    }

    // ---- park pools: entry e of field f sits at base + (f * K + e) * 16 (fields 0-4) / base + 80 * K + e * 8 (field 5), K = the pool's capacity ----
    RTD_INLINE void park_store(unsigned char *base, uint32_t K, uint32_t e) {
        d2 *f = (d2 *) base;
        d2 v;
        v.x = o.x; v.y = o.y; f[e] = v;
        v.x = o.z; v.y = d.x; f[K + e] = v;
        v.x = d.y; v.y = d.z; f[2u * K + e] = v;
        i4 r; r.x = __double2loint(w.bestLen); r.y = __double2hiint(w.bestLen); r.z = (int) colour; r.w = (int) slotOff;
        ((i4 *) base)[3u * K + e] = r;
        r.x = (int) rng.x; r.y = (int) rng.y; r.z = (int) rng.z; r.w = (int) rng.w;
        ((i4 *) base)[4u * K + e] = r;
        i2 b; b.x = bounces; b.y = w.best;
        ((i2 *) (base + 80u * K))[e] = b;
    }
    RTD_INLINE void park_load(const unsigned char *base, uint32_t K, uint32_t e, int state) {
        const d2 *f = (const d2 *) base;
        const d2 a = f[e], b = f[K + e], c = f[2u * K + e];
        const i4 t = ((const i4 *) base)[3u * K + e], r = ((const i4 *) base)[4u * K + e];
        const i2 bo = ((const i2 *) (base + 80u * K))[e];
        o = mk(a.x, a.y, b.x); d = mk(b.y, c.x, c.y);
        w.bestLen = __hiloint2double(t.y, t.x); colour = (uint32_t) t.z; slotOff = (uint32_t) t.w;
        rng.x = (uint32_t) r.x; rng.y = (uint32_t) r.y; rng.z = (uint32_t) r.z; rng.w = (uint32_t) r.w;
        bounces = bo.x; w.best = bo.y;
        w.off = end;
        st = state;
    }
    RTD_INLINE unsigned char *pool_l() const { return pool + (size_t) RTD_PARK_ENTRY_BYTES * (size_t) p.park; }
    RTD_INLINE unsigned char *pool_t() const { return pool + (size_t) RTD_PARK_ENTRY_BYTES * (size_t) (p.park + (p.park_l_lds ? 0 : p.park_l)); } // capacity p.park
    // A parked LAMBERT hit is {strike (in o), inside (bit 31 of bounces), colour, rng, slot, bounces, object}: what lambert_bounce needs.
    // In LDS: entry e of field f at base + (f * K + e) * 16 (fields 0-2), base + 48 * K + e * 8 (field 3).
    RTD_INLINE void park_store_lds(uint32_t K, uint32_t e) {
        RTD_AS3 d2 *f = (RTD_AS3 d2 *) poolLds;
        d2 v; v.x = o.x; v.y = o.y; f[e] = v;
        i4 r; r.x = __double2loint(o.z); r.y = __double2hiint(o.z); r.z = (int) colour; r.w = (int) slotOff;
        ((RTD_AS3 i4 *) poolLds)[K + e] = r;
        r.x = (int) rng.x; r.y = (int) rng.y; r.z = (int) rng.z; r.w = (int) rng.w;
        ((RTD_AS3 i4 *) poolLds)[2u * K + e] = r;
        i2 b; b.x = bounces; b.y = w.best;
        ((RTD_AS3 i2 *) (poolLds + 48u * K))[e] = b;
    }
    RTD_INLINE void park_load_lds(uint32_t K, uint32_t e) {
        const d2 a = ((const RTD_AS3 d2 *) poolLds)[e];
        const i4 t = ((const RTD_AS3 i4 *) poolLds)[K + e], r = ((const RTD_AS3 i4 *) poolLds)[2u * K + e];
        const i2 bo = ((const RTD_AS3 i2 *) (poolLds + 48u * K))[e];
        o = mk(a.x, a.y, __hiloint2double(t.y, t.x)); colour = (uint32_t) t.z; slotOff = (uint32_t) t.w;
        rng.x = (uint32_t) r.x; rng.y = (uint32_t) r.y; rng.z = (uint32_t) r.z; rng.w = (uint32_t) r.w;
        bounces = bo.x; w.best = bo.y;
        w.off = end;
        st = L_LAMB;
    }
    // How the `nIdle` idle lanes of a refill are served (wave-uniform): a FULL batch of parked Lambert hits if that pool holds one,
    // else a full batch of parked general hits, else new items; once there are no new items, whatever the pools still hold.
    RTD_INLINE void unpark_plan(uint32_t nIdle, bool haveNew, uint32_t &nUnL, uint32_t &nUnA, uint32_t &nUnT) const {
        nUnL = nUnA = nUnT = 0u;
        if (parkedL >= nIdle) nUnL = nIdle;
        else if (parked >= nIdle) nUnA = nIdle;
        else if (TEX && parkedT >= nIdle) nUnT = nIdle;
        else if (!haveNew) {
            nUnL = parkedL;
            nUnA = parked < nIdle - nUnL ? parked : nIdle - nUnL;
            if (TEX) nUnT = parkedT < nIdle - nUnL - nUnA ? parkedT : nIdle - nUnL - nUnA;
        }
    }
    // the idle lane of rank `rank` takes its parked path, if the plan gives it one
    RTD_INLINE bool unpark_lane(uint32_t rank, uint32_t nUnL, uint32_t nUnA, uint32_t nUnT) {
        // (the three counters are read HERE, unconditionally: read inside the branches, the optimiser merges the loads into one load
        // from a selected address, and the scheduler's state then lives in scratch memory instead of registers)
        const uint32_t topL = parkedL - 1u, topA = parked - 1u, topT = parkedT - 1u;
        if (rank < nUnL) {
            if (p.park_l_lds) park_load_lds((uint32_t) p.park_l, topL - rank);
            else park_load(pool_l(), (uint32_t) p.park_l, topL - rank, L_LAMB);
            return true;
        }
        if (rank < nUnL + nUnA) { park_load(pool, (uint32_t) p.park, topA - (rank - nUnL), L_SLOW); texc = RTD_NO_TEX; return true; }
        if (TEX && rank < nUnL + nUnA + nUnT) { park_load(pool_t(), (uint32_t) p.park, topT - (rank - nUnL - nUnA), L_TEX); return true; }
        return false;
    }

    // what follows Hittable.Reflection in Scene.traceRay (Scene.fs:105-112)
    RTD_INLINE void after_reflection(bool absorbed) {
        bool done = absorbed;
        uint32_t res = colour;
        if (!absorbed) {
            bounces = bounces + 1;
            if (bounces > p.depth) { done = true; res = RTD_HOTPINK; } // Scene.fs:98,114
        }
        if (done) { ended = true; result = res; st = L_IDLE; w.off = end; }
        else {
            st = L_WALK;
            walk_begin(w, sc.first);
            if (COUNT) cnt.rays++;
            log_ray();
        }
    }

    // ---- slow: the general Hittable.Reflection for the lanes whose hit is not one of the two common cases ----
    RTD_INLINE void stage_slow() {
        const unsigned long long m = __builtin_amdgcn_ballot_w64(st == L_SLOW);
        if (m == 0ull) return;
        if (COUNT || RTD_CLK) { ss.slow++; ss.slowLanes += (uint32_t) __popcll(m); }
        if (st == L_SLOW) {
            const V3 strike = walk(o, d, w.bestLen); // Ray.walkAlong ray bestLength (Scene.fs:91)
            after_reflection(reflection<LDS, false>(sc, w.best, strike, o, d, colour, rng, TEX ? texc : RTD_NO_TEX));
        }
    }

    // ---- tex: Texture.colourAt (Texture.fs:50-67) for the lanes that took parked textured hits (or whose hit found that pool full).
    // The evaluation is long (correctly rounded acos / atan2 / sin in double-double, rt_trig.h); here it runs for a batch of lanes
    // at once, inline, and nowhere else in the kernel.  The lane goes on to the general reflection with the colour in `texc`.
    RTD_INLINE void stage_tex() {
        if (!TEX) return;
        if (__builtin_amdgcn_ballot_w64(st == L_TEX) == 0ull) return;
        if (st == L_TEX) {
            const V3 strike = walk(o, d, w.bestLen);
            texc = texture_colour_at_inline(sc.tex, sc.texels, textured(sc.meta[w.best]), strike, nullptr);
            st = L_SLOW;
        }
    }

    // ---- lamb: the Lambert bounce (Sphere.fs:202-222) for the lanes that took parked Lambert hits (or whose hit found the pool full) ----
    RTD_INLINE void stage_lamb() {
        if (__builtin_amdgcn_ballot_w64(st == L_LAMB) == 0ull) return;
        const unsigned long long k0 = RTD_CLK ? __builtin_amdgcn_s_memtime() : 0ull;
        if (st == L_LAMB) { // o holds the strike point, bit 31 of bounces "the ray came from inside" (stage_shade)
            const bool inside = bounces < 0;
            bounces &= 0x7FFFFFFF;
            lambert_bounce<LDS>(sc, w.best, sc.meta[w.best], o, inside, o, d, colour, rng);
            after_reflection(false);
        }
        if (RTD_CLK) ss.tLamb += __builtin_amdgcn_s_memtime() - k0;
    }

    // ---- walk: BoundingBox.hits over the tree image, leaf tests deferred out of the node loop ----
    // A lane steps iff w.off < end: finished walks sit at >= end, pending leaves carry RTD_LEAF (> end), idle lanes are
    // parked at `end`.  So one compare gives the active set, and waiting = busy - active.
    RTD_INLINE void stage_walk() {
        if (__builtin_amdgcn_ballot_w64(st == L_WALK) == 0ull) return;
        const int nBusy = __popcll(__builtin_amdgcn_ballot_w64(st != L_IDLE));
        const int stop = (nBusy - p.leaf_wait) > 0 ? (nBusy - p.leaf_wait) : 0; // active <= stop  <=>  waiting >= leaf_wait
        if constexpr (!COUNT) {
            // the timed variant: the hand-written single-precision filter loop with its queue of pending leaves (rt_device.h) over the
            // LDS copy of the scene, or over global memory for a scene that does not fit; the leaf pass makes the sphere test and,
            // where the hit does not imply it, the leaf's exact box test; a lane is finished when its walk is exhausted AND its
            // queue is empty
            const WalkCtx32 f = walk_ctx32(o, d, p.off.bmax);
            double bestF = (w.best < 0) ? __builtin_inf() : w.bestLen * w.bestLen; // `a = point * point` (Scene.fs:45), recomputed
            const bool implied = p.off.box_implied != 0; // (rays of this kernel are unitised: Ray.make')
            if (RTD_CLK) ss.trips++; // (diagnostic build: walk-stage entries and leaf passes; the loop's trips are not counted)
            for (;;) {
                const unsigned long long k0 = RTD_CLK ? __builtin_amdgcn_s_memtime() : 0ull;
#ifdef RTD_STAGE_CLOCKS
                unsigned nTrips = 0u, nLanes = 0u;
#define RTD_TRIP_PASS , nTrips, nLanes
#else
#define RTD_TRIP_PASS
#endif
                if constexpr (LDS) w.off = node_loop_lds32(w.off, pend, end, stop, f RTD_TRIP_PASS);
                else if (sc.narrow) w.off = node_loop_hyb16(w.off, pend, (const unsigned char *) sc.node, sc.lds_lim, sc.lds_thr, end, stop, f RTD_TRIP_PASS);
                else w.off = node_loop_glb32(w.off, pend, pend1, (const unsigned char *) sc.node, sc.lds_lim, sc.lds_thr, end, stop, f);
#ifdef RTD_STAGE_CLOCKS
                ss.loopTrips += nTrips; ss.loopLanes += nLanes;
#endif
#undef RTD_TRIP_PASS
                const unsigned long long k1 = RTD_CLK ? __builtin_amdgcn_s_memtime() : 0ull;
                if (RTD_CLK && __builtin_amdgcn_ballot_w64(pend != 0u) != 0ull) ss.leaf++;
                if (pend != 0u) {
                    int prim;
                    if (LDS || sc.narrow) {
                        prim = pend_pop(pend);
                        if (pend == 0u) { pend = pend1; pend1 = 0u; } // a camera ray's third and fourth candidate (start_item)
                    } else prim = pend_pop_wide(pend, pend1);
                    leaf_test_object_exact<LDS>(sc, o, d, bestF, w, prim, implied);
                }
                if (RTD_CLK) { ss.tLoop += k1 - k0; ss.tLeaf += __builtin_amdgcn_s_memtime() - k1; }
                const bool fin = (st == L_WALK) && (w.off >= end) && pend == 0u;
                if (fin) st = L_DONE;
                const int nWalk = __popcll(__builtin_amdgcn_ballot_w64(st == L_WALK));
                const int nDone = __popcll(__builtin_amdgcn_ballot_w64(st == L_DONE));
                if (nWalk == 0 || nDone >= p.yield_lanes) break;
            }
            return;
        }
        WalkCtx c = walk_ctx(d, w); // cheap to re-derive; keeps 9 doubles out of the parked state
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
            const bool fin = (st == L_WALK) && (w.off >= end);
            if (fin) st = L_DONE;
            const int nWalk = __popcll(__builtin_amdgcn_ballot_w64(st == L_WALK));
            const int nDone = __popcll(__builtin_amdgcn_ballot_w64(st == L_DONE));
            if (nWalk == 0 || nDone >= p.yield_lanes) break;
        }
    }

    // ---- shade: the rest of Scene.hitObject (Scene.fs:77-91), then Hittable.Reflection for the common cases; the others park ----
    RTD_INLINE void stage_shade() {
        const unsigned long long dm = __builtin_amdgcn_ballot_w64(st == L_DONE);
        if (dm == 0ull) return;
        if (COUNT || RTD_CLK) { ss.shade++; ss.shadeLanes += (uint32_t) __popcll(dm); }
        if (st == L_DONE) {
            const unsigned long long k0 = RTD_CLK ? __builtin_amdgcn_s_memtime() : 0ull;
            unbounded_tests<LDS, COUNT>(sc, o, d, w, cnt);
            if (RTD_CLK) ss.tUnb += __builtin_amdgcn_s_memtime() - k0;
            if (w.best < 0) { ended = true; result = RTD_BLACK; st = L_IDLE; w.off = end; } // "never heard from again": Black (Scene.fs:102-104)
            else {
                if (COUNT) cnt.refl++;
                const i2 m = sc.meta[w.best];
                if (fast_style(m)) {
                    if (p.park_l > 0 && (((uint32_t) m.x >> 2) & 7u) != 0u) {
                        // an untextured Lambert sphere: parked as {strike, inside} and bounced in full batches (stage_lamb)
                        const bool inside = lambert_inside<LDS>(sc, w.best, m, o);
                        o = walk(o, d, w.bestLen); // Ray.walkAlong ray bestLength (Scene.fs:91)
                        bounces |= inside ? (int) 0x80000000u : 0;
                        st = L_LAMB;
                    } else {
                        const V3 strike = walk(o, d, w.bestLen); // Ray.walkAlong ray bestLength (Scene.fs:91)
                        after_reflection(reflection_fast<LDS>(sc, w.best, m, strike, o, d, colour, rng));
                    }
                } else if (TEX && textured(m) >= 0) st = L_TEX;
                else { st = L_SLOW; texc = RTD_NO_TEX; }
            }
        }
        if (TEX && p.park > 0) {
            const unsigned long long tm = __builtin_amdgcn_ballot_w64(st == L_TEX);
            if (tm != 0ull) {
                const uint32_t room = (uint32_t) p.park - parkedT, want = (uint32_t) __popcll(tm);
                const uint32_t rank = lane_rank(tm);
                if (st == L_TEX && rank < room) { park_store(pool_t(), (uint32_t) p.park, parkedT + rank); st = L_IDLE; w.off = end; }
                parkedT += want < room ? want : room;
            }
        }
        if (p.park > 0) {
            const unsigned long long sm = __builtin_amdgcn_ballot_w64(st == L_SLOW);
            if (sm != 0ull) {
                const uint32_t room = (uint32_t) p.park - parked, want = (uint32_t) __popcll(sm);
                const uint32_t rank = lane_rank(sm);
                if (st == L_SLOW && rank < room) { park_store(pool, (uint32_t) p.park, parked + rank); st = L_IDLE; w.off = end; }
                const uint32_t n = want < room ? want : room;
                parked += n;
                if (COUNT || RTD_CLK) ss.parkedLanes += n;
            }
        }
        if (p.park_l > 0) {
            const unsigned long long lm = __builtin_amdgcn_ballot_w64(st == L_LAMB);
            if (lm != 0ull) {
                const uint32_t room = (uint32_t) p.park_l - parkedL, want = (uint32_t) __popcll(lm);
                const uint32_t rank = lane_rank(lm);
                if (st == L_LAMB && rank < room) {
                    if (p.park_l_lds) park_store_lds((uint32_t) p.park_l, parkedL + rank);
                    else park_store(pool_l(), (uint32_t) p.park_l, parkedL + rank);
                    st = L_IDLE; w.off = end;
                }
                parkedL += want < room ? want : room;
            }
        }
    }

    // PixelStats.add (Pixel.fs:97-101) for a lane whose path ended this turn; Count is implied by the item count
    RTD_INLINE void add_result(RTD_AS3 uint32_t *acc3) const {
        if (result != 0u) {
            lds_add(acc3 + 0, result & 0xFFu);
            lds_add(acc3 + 1, (result >> 8) & 0xFFu);
            lds_add(acc3 + 2, (result >> 16) & 0xFFu);
        }
    }
};

// Trace `total` items of the current unit.  Item i belongs to pixel slot map[i / per] (or i / per when map is null)
// and is sample s_base + i % per of that pixel; its colour is added to accumulator slot (sample < split ? 0 : 1).
template <bool LDS, bool COUNT, bool COST, bool TEX, bool LOG>
RTD_INLINE void run_items(const RenderParams &p, const SceneView<LDS> &sc, unsigned char *pool, RTD_AS3 unsigned char *poolLds, RTD_AS3 uint32_t *acc, const RTD_AS3 uint32_t *pix,
                          const RTD_AS3 uint32_t *cand, const RTD_AS3 uint32_t *live, bool use_live, uint32_t total, uint32_t per, uint32_t s_base,
                          uint32_t split, Counters &cnt, StageStats &ss) {
    Sched<LDS, COUNT, TEX, LOG> L(p, sc, cnt, ss, pool, poolLds);
    uint32_t next = 0; // wave-uniform
    const bool fastDiv = total < (1u << 22) && per < (1u << 23); // see div_uniform
    const float perRcp = 1.0f / (float) per;
    for (;;) {
        L.ended = false;
        const unsigned long long t0 = (COUNT || RTD_CLK) ? __builtin_amdgcn_s_memtime() : 0ull;
        // ---- refill: idle lanes take parked paths or the next items of the unit ----
        const unsigned long long idle = __builtin_amdgcn_ballot_w64(L.st == L_IDLE);
        const bool haveNew = next < total;
        if (idle != 0ull && (haveNew || (L.parked | L.parkedL | L.parkedT) != 0u) && (__popcll(idle) >= p.refill_lanes || ~idle == 0ull)) {
            const uint32_t nIdle = (uint32_t) __popcll(idle);
            const uint32_t rank = lane_rank(idle);
            uint32_t nUnL, nUnA, nUnT;
            L.unpark_plan(nIdle, haveNew, nUnL, nUnA, nUnT);
            const uint32_t nUn = nUnL + nUnA + nUnT;
            if (COUNT || RTD_CLK) { ss.refill++; ss.refillLanes += nIdle; }
            if (L.st == L_IDLE) {
                if (!L.unpark_lane(rank, nUnL, nUnA, nUnT)) {
                    const uint32_t item = next + (rank - nUn);
                    if (item < total) {
                        uint32_t j = fastDiv ? div_uniform(item, per, perRcp) : item / per;
                        uint32_t s = s_base + (item - j * per);
                        uint32_t slot = use_live ? live[j] : j;
                        int row = (int) pix[slot * 4 + 0], col = (int) pix[slot * 4 + 1];
                        uint64_t pkey = (uint64_t) pix[slot * 4 + 2] | ((uint64_t) pix[slot * 4 + 3] << 32);
                        // low half: acc word, high half: pixel slot
                        L.start_item(pkey, s, row, col, (((s < split) ? 0u : (uint32_t) p.chunk * 3u) + slot * 3u) | (slot << 16), cand[slot * 2], cand[slot * 2 + 1]);
                    }
                }
            }
            L.parked -= nUnA;
            L.parkedL -= nUnL;
            L.parkedT -= nUnT;
            next += nIdle - nUn;
            next = __builtin_amdgcn_readfirstlane(next);
        }
        if (__builtin_amdgcn_ballot_w64(L.st != L_IDLE) == 0ull) {
            if (next >= total && (L.parked | L.parkedL | L.parkedT) == 0u) break;
            continue;
        }
        const unsigned long long t1 = (COUNT || RTD_CLK) ? __builtin_amdgcn_s_memtime() : 0ull;
        L.stage_tex();
        L.stage_slow();
        L.stage_lamb();
        const unsigned long long t2 = (COUNT || RTD_CLK) ? __builtin_amdgcn_s_memtime() : 0ull;
        L.stage_walk();
        const unsigned long long t3 = (COUNT || RTD_CLK) ? __builtin_amdgcn_s_memtime() : 0ull;
        L.stage_shade();
        if (COUNT || RTD_CLK) { const unsigned long long t4 = __builtin_amdgcn_s_memtime(); ss.tRefill += t1 - t0; ss.tSlow += t2 - t1; ss.tWalk += t3 - t2; ss.tShade += t4 - t3; }
        if (L.ended) {
            L.add_result(acc + (L.slotOff & 0xFFFFu));
            if (COST) lds_add(acc + 10u * (uint32_t) p.chunk + (L.slotOff >> 16), (uint32_t) L.bounces + 1u); // ~ Scene.hitObject calls of this path
        }
    }
}

// Pass B (MODE 2): phase 2 for the pixels of the cost-ordered list, STREAMED.  A wave reserves a run of list entries ("range"),
// hands out its npx*n2 items, and while the last paths of that range are still in flight it already reserves the next range and
// hands out its items: two accumulator slots alternate, a range is flushed (its sums added to what pass A left in `accum`) when
// its last path has ended.  There is no dependency between ranges, so no lane waits at a range boundary -- which is what makes
// small ranges (good load balance across waves) affordable.  Lane states and stage scheduling are Sched's.
template <bool LDS, bool COUNT, bool TEX>
RTD_INLINE void run_stream(const RenderParams &p, const SceneView<LDS> &sc, unsigned char *pool, RTD_AS3 unsigned char *poolLds, RTD_AS3 uint32_t *wv, uint32_t n1, uint32_t n2, Counters &cnt,
                           StageStats &ss, uint64_t &sampleCount) {
    const int lane = threadIdx.x & 63;
    const uint32_t P = (uint32_t) p.chunk;
    const uint32_t SW = 7u * P; // words per slot: acc [P][3] then pix [P][4]
    const unsigned long long nList = (unsigned long long) *p.live_count;
    Sched<LDS, COUNT, TEX, false> L(p, sc, cnt, ss, pool, poolLds); // slotOff: word offset from wv of the path's accumulator triple (>= SW: slot 1)

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
        L.ended = false;
        const unsigned long long t0 = (COUNT || RTD_CLK) ? __builtin_amdgcn_s_memtime() : 0ull;
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
                        const CameraParams *cp = p.cam_ptr;
                        asm volatile("" : "+s"(cp));
                        uint32_t c2;
                        const uint32_t c1 = pixel_candidates<LDS, !COUNT>(sc, *cp, p.max_h - (int) r - 1, (int) c - p.max_w, c2);
                        wv[14u * P + (curSlot * P + lane) * 2u] = c1;
                        wv[14u * P + (curSlot * P + lane) * 2u + 1u] = c2;
                    }
                    __builtin_amdgcn_wave_barrier();
                    curFirst = first; curNpx = npx; curNext = 0u; curTotal = npx * n2; curOut = 0u;
                }
            }
        }
        const unsigned long long idle = __builtin_amdgcn_ballot_w64(L.st == L_IDLE);
        if (curNpx == 0u && prevNpx == 0u && exhausted && idle == ~0ull) break; // parked paths keep their range open, so none is left

        // ---- refill: parked paths, or items of the current range ----
        {
            const uint32_t nIdle = (uint32_t) __popcll(idle);
            const uint32_t avail = curTotal - curNext;
            if (nIdle != 0u && (avail != 0u || (L.parked | L.parkedL | L.parkedT) != 0u) && ((int) nIdle >= p.refill_lanes || nIdle == 64u)) {
                const uint32_t rank = lane_rank(idle);
                uint32_t nUnL, nUnA, nUnT;
                L.unpark_plan(nIdle, avail != 0u, nUnL, nUnA, nUnT);
                const uint32_t nUn = nUnL + nUnA + nUnT;
                if (COUNT || RTD_CLK) { ss.refill++; ss.refillLanes += nIdle; }
                const uint32_t rest = nIdle - nUn;
                const uint32_t take = rest < avail ? rest : avail;
                bool started = false;
                if (L.st == L_IDLE) {
                    if (!L.unpark_lane(rank, nUnL, nUnA, nUnT) && rank - nUn < take) {
                        const uint32_t item = curNext + (rank - nUn);
                        const uint32_t j = fastDiv ? div_uniform(item, n2, perRcp) : item / n2;
                        const uint32_t smp = n1 + (item - j * n2);
                        const RTD_AS3 uint32_t *pix = wv + curSlot * SW + 3u * P;
                        const int row = (int) pix[j * 4 + 0], col = (int) pix[j * 4 + 1];
                        const uint64_t pkey = (uint64_t) pix[j * 4 + 2] | ((uint64_t) pix[j * 4 + 3] << 32);
                        started = L.start_item(pkey, smp, row, col, curSlot * SW + j * 3u, wv[14u * P + (curSlot * P + j) * 2u], wv[14u * P + (curSlot * P + j) * 2u + 1u]);
                    }
                }
                L.parked -= nUnA;
                L.parkedL -= nUnL;
                L.parkedT -= nUnT;
                curNext += take;
                curOut += (uint32_t) __popcll(__builtin_amdgcn_ballot_w64(started));
            }
        }
        if (__builtin_amdgcn_ballot_w64(L.st != L_IDLE) == 0ull) continue;

        const unsigned long long t1 = (COUNT || RTD_CLK) ? __builtin_amdgcn_s_memtime() : 0ull;
        L.stage_tex();
        L.stage_slow();
        L.stage_lamb();
        const unsigned long long t2 = (COUNT || RTD_CLK) ? __builtin_amdgcn_s_memtime() : 0ull;
        L.stage_walk();
        const unsigned long long t3 = (COUNT || RTD_CLK) ? __builtin_amdgcn_s_memtime() : 0ull;
        L.stage_shade();
        if (COUNT || RTD_CLK) { const unsigned long long t4 = __builtin_amdgcn_s_memtime(); ss.tRefill += t1 - t0; ss.tSlow += t2 - t1; ss.tWalk += t3 - t2; ss.tShade += t4 - t3; }
        if (L.ended) L.add_result(wv + L.slotOff);
        const bool inCur = (curNpx != 0u) && ((L.slotOff >= SW) == (curSlot == 1u));
        curOut -= (uint32_t) __popcll(__builtin_amdgcn_ballot_w64(L.ended && inCur));
        prevOut -= (uint32_t) __popcll(__builtin_amdgcn_ballot_w64(L.ended && !inCur));
    }
}

// Stages the LDS part of the scene image into the workgroup's LDS, 16 B per lane per trip, coalesced, and makes the node links
// absolute LDS addresses: a walk position then IS the record's address (no add per visit).  Returns the bytes used.
// F32 = false (the counting variant): node | geo | meta, the exact double-precision records, walked by the compiled node_step.
// F32 = true (the timed variant): geo | meta | node32, the single-precision filter records in their queue form (rt_scene.h),
// walked by node_loop_lds32.
template <int BLOCK, bool F32>
RTD_INLINE uint32_t stage_scene(const RenderParams &p, unsigned char *smem) {
    const uint32_t sceneBytes = F32 ? p.off.lds32_total : p.off.lds_total;
    const d2 *src = (const d2 *) (p.scene_image + (F32 ? p.off.geo : 0u));
    RTD_AS3 d2 *dst = (RTD_AS3 d2 *) smem;
    for (uint32_t i = threadIdx.x; i < sceneBytes / 16u; i += BLOCK) dst[i] = src[i];
    __syncthreads();
    RTD_AS3 unsigned char *nodes = (RTD_AS3 unsigned char *) smem + (F32 ? p.off.node32 - p.off.geo : p.off.node);
    const int first = (int) (uint32_t) (uintptr_t) nodes;
    for (int i = threadIdx.x; i < p.off.n_nodes; i += BLOCK) {
        if (F32) {
            RTD_AS3 i2 *lk = (RTD_AS3 i2 *) (nodes + i * RTD_NODE32_BYTES + 48);
            i2 v = *lk; // on_hit, on_miss
            v.x += first;
            v.y += first;
            *lk = v;
        } else {
            RTD_AS3 i2 *lk = (RTD_AS3 i2 *) (nodes + i * RTD_NODE_BYTES + 96);
            i2 v = *lk;
            v.x += first; // a Leaf's RTD_LEAF flag (bit 30) is above every LDS address and survives the add
            v.y += first;
            *lk = v;
        }
    }
    __syncthreads();
    return sceneBytes;
}

// "Hybrid" (the timed variant of a scene that does not fit the LDS): the first p.lds_node_bytes of the node32 section -- its records
// are ordered by depth -- go to the START of the workgroup's LDS, links untouched (byte offsets from the section's start), so that a
// walk position below that limit is an LDS address as it is (node_loop_glb32).  Needs the dynamic LDS to start at address 0;
// returns the bytes staged, 0 (nothing in LDS, every visit from global memory) if it does not.
template <int BLOCK>
RTD_INLINE uint32_t stage_nodes32(const RenderParams &p, unsigned char *smem) {
    if ((uint32_t) (uintptr_t) (RTD_AS3 unsigned char *) smem != 0u) return 0u; // (the reservation stays unused; the host cannot know)
    const uint32_t bytes = (uint32_t) p.lds_node_bytes;
    const d2 *src = (const d2 *) (p.scene_image + p.off.node32);
    RTD_AS3 d2 *dst = (RTD_AS3 d2 *) smem;
    for (uint32_t i = threadIdx.x; i < bytes / 16u; i += BLOCK) dst[i] = src[i];
    __syncthreads();
    return bytes;
}

// MODE 0: fused -- a unit's pixels go through phase 1, the adaptive decision and phase 2 on one wave.
// MODE 1: pass A -- phase 1 and the decision for every pixel; pixels that continue are appended to `pairs` with the number of
//         rays their 2k+1 samples took (a cost estimate), the others are final.
// MODE 2: pass B -- phase 2 for the pixels of `live_list`, which the host-side launch sequence has ordered by decreasing cost
//         (longest job first); run_stream.
// MODE 3: MODE 0 with the ray log of rt_scene_tune's probe compiled in (a dozen more scalar values live across the loop: kept out
//         of the kernels that render frames).
// Per-pixel cost is heavy-tailed (a pixel on a glass sphere: ~20 rays per sample, 4 ms of one wave), so when a shard has only a few
// units per wave the fused kernel ends with most waves waiting for a few long units started late; A + sort + B removes that tail.
// Every mode computes the same integers: which wave traces which sample when has no effect (streams are per item).
// TEX: the scene has parameterised textures (see `reflection`).
template <bool LDS, bool COUNT, int BLOCK, int MODE, bool TEX>
__global__ void __launch_bounds__(BLOCK) render_kernel(const RenderParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr bool FUSED = MODE == 0 || MODE == 3;
    const int lane = threadIdx.x & 63;
    // the wave's index as a SCALAR: everything derived from it (the wave's LDS scratch, its park pools) then has a scalar base, and the
    // pools' field addresses are scalar base + 32-bit lane offset instead of 64-bit vector arithmetic kept alive across the loop
    const int wave = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));

    // LDS part: the whole scene (LDS variants), or as much of the tree as fits (the timed variant of a larger scene), or nothing
    const uint32_t ldsNodes = (!LDS && !COUNT) ? stage_nodes32<BLOCK>(p, smem) : 0u;
    const uint32_t sceneBytes = LDS ? stage_scene<BLOCK, !COUNT>(p, smem) : (uint32_t) ((!LDS && !COUNT) ? p.lds_node_bytes : 0);
    SceneView<LDS> scv = make_view<LDS, !COUNT>(p, smem);
    scv.lds_lim = (int) ldsNodes;
    scv.lds_thr = p.lds_node_thr;
    const SceneView<LDS> &sc = scv;
    const uint32_t P = (uint32_t) p.chunk;
    RTD_AS3 uint32_t *wv = (RTD_AS3 uint32_t *) (smem + sceneBytes) + (size_t) wave * (MODE == 1 ? RTD_WAVE_WORDS_A(P) : RTD_WAVE_WORDS(P));
    RTD_AS3 uint32_t *acc = wv;
    RTD_AS3 uint32_t *pix = wv + 6u * P;
    RTD_AS3 uint32_t *live = pix + 4u * P;
    RTD_AS3 uint32_t *cand = live + P; // (fused and pass A; pass B keeps its own behind the two slots)
    unsigned char *pool = p.park_pool + ((size_t) blockIdx.x * (BLOCK / 64) + (size_t) wave) * (size_t) RTD_PARK_ENTRY_BYTES *
                                        (size_t) (p.park + (p.park_l_lds ? 0 : p.park_l) + (TEX ? p.park : 0));
    // the Lambert pools in LDS (if any) follow the waves' scratch
    RTD_AS3 unsigned char *poolLds = (RTD_AS3 unsigned char *) (smem + sceneBytes) + (size_t) (BLOCK / 64) * (MODE == 1 ? RTD_WAVE_WORDS_A(P) : RTD_WAVE_WORDS(P)) * 4u +
                                     (size_t) wave * (size_t) RTD_PARK_L_LDS_BYTES * (size_t) p.park_l;

    const uint64_t nLocal = (uint64_t) p.n_rows * (uint64_t) p.cols;
    const uint32_t k = (uint32_t) p.k;
    const uint32_t n1 = 2u * k + 1u;
    const int n2s = p.spp - 2 * p.k - 1; // Scene.fs:191
    const uint32_t n2 = n2s > 0 ? (uint32_t) n2s : 0u;

    const unsigned long long tStart = COUNT ? __builtin_amdgcn_s_memrealtime() : 0ull;
    Counters cnt; cnt.rays = cnt.aabb = cnt.prim = cnt.refl = 0;
    StageStats ss; ss.refill = ss.trips = ss.leaf = ss.shade = ss.refillLanes = ss.shadeLanes = ss.slow = ss.slowLanes = ss.parkedLanes = 0;
    ss.tRefill = ss.tSlow = ss.tWalk = ss.tShade = 0ull;
    ss.tLoop = ss.tLeaf = ss.tUnb = ss.tCam = ss.tLamb = 0ull;
    ss.loopTrips = ss.loopLanes = 0ull;
    uint32_t earlyCount = 0;
    uint64_t sampleCount = 0; // Scene.traceOnce calls = sum of PixelStats.Count

    if (MODE == 2) run_stream<LDS, COUNT, TEX>(p, sc, pool, poolLds, wv, n1, n2, cnt, ss, sampleCount);
    else
    for (;;) {
        uint32_t unit = 0;
        if (lane == 0) unit = atomicAdd(p.queue, 1u);
        unit = __builtin_amdgcn_readfirstlane(unit);
        const unsigned long long first = (unsigned long long) unit * P;
        if (first >= nLocal) break;
        const uint32_t npx = (uint32_t) ((nLocal - first < (unsigned long long) P) ? (nLocal - first) : (unsigned long long) P);

        // per-pixel coordinates (Scene.fs:219,226) and stream key; clear the accumulators
        for (uint32_t i = (uint32_t) lane; i < 6u * P; i += 64u) acc[i] = 0u;
        if (MODE == 1 && (uint32_t) lane < P) acc[10u * P + lane] = 0u;
        unsigned long long lp = 0; // local pixel of lane j < npx
        if ((uint32_t) lane < npx) {
            lp = first + (uint32_t) lane;
            uint32_t lr = (uint32_t) (lp / (unsigned long long) p.cols);
            uint32_t c = (uint32_t) (lp - (unsigned long long) lr * (unsigned long long) p.cols);
            uint32_t r = (uint32_t) p.row_first + lr * (uint32_t) p.row_stride;
            uint64_t pkey = pixel_key(p.seed_key, (uint64_t) r * (uint64_t) p.cols + c); // global pixel index
            pix[lane * 4 + 0] = (uint32_t) (p.max_h - (int) r - 1);
            pix[lane * 4 + 1] = (uint32_t) ((int) c - p.max_w);
            pix[lane * 4 + 2] = (uint32_t) pkey;
            pix[lane * 4 + 3] = (uint32_t) (pkey >> 32);
            const CameraParams *cp = p.cam_ptr;
            asm volatile("" : "+s"(cp));
            uint32_t c2;
            const uint32_t c1 = pixel_candidates<LDS, !COUNT>(sc, *cp, p.max_h - (int) r - 1, (int) c - p.max_w, c2);
            cand[lane * 2] = c1;
            cand[lane * 2 + 1] = c2;
        }
        __builtin_amdgcn_wave_barrier();

        // ---- phase 1: 2k+1 samples per pixel, sums split after sample k (Scene.fs:172-182) ----
        run_items<LDS, COUNT, MODE == 1, TEX, MODE == 3>(p, sc, pool, poolLds, acc, pix, cand, live, false, npx * n1, n1, 0u, k + 1u, cnt, ss);
        __builtin_amdgcn_wave_barrier();

        // ---- decide (Scene.fs:177-188) and compact the pixels that continue ----
        int sumR = 0, sumG = 0, sumB = 0, count = 0;
        bool cont = false;
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
        const uint32_t nLive = (uint32_t) __popcll(liveMask);
        const uint32_t pos = lane_rank(liveMask);
        if (FUSED) {
            if (cont) live[pos] = (uint32_t) lane;
        } else if (nLive > 0u) { // pass A: hand the pixel over to pass B, with its cost estimate
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(p.live_count, nLive);
            base = __builtin_amdgcn_readfirstlane(base);
            if (cont) p.pairs[base + pos] = ((unsigned long long) acc[10u * P + lane] << 32) | (unsigned long long) (uint32_t) lp;
        }
        __builtin_amdgcn_wave_barrier();

        // ---- phase 2: the remaining spp-2k-1 samples of the surviving pixels (Scene.fs:191-192) ----
        if (FUSED && nLive > 0u) {
            run_items<LDS, COUNT, false, TEX, MODE == 3>(p, sc, pool, poolLds, acc, pix, cand, live, true, nLive * n2, n2, n1, 0xFFFFFFFFu, cnt, ss);
            __builtin_amdgcn_wave_barrier();
        }

        // ---- PixelStats and mean out: one 16-byte store per pixel ----
        if ((uint32_t) lane < npx) {
            if (FUSED && cont) {
                sumR += (int) lds_take(acc + lane * 3 + 0);
                sumG += (int) lds_take(acc + lane * 3 + 1);
                sumB += (int) lds_take(acc + lane * 3 + 2);
                count += (int) n2;
            }
            sampleCount += (uint64_t) count;
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

    // ---- counters: one wave reduction and a few atomics per wave per launch ----
    uint64_t e = wave_sum_u64(earlyCount), s = wave_sum_u64(sampleCount);
    if (COUNT) {
        uint64_t a = wave_sum_u64(cnt.rays), b = wave_sum_u64(cnt.aabb), c = wave_sum_u64(cnt.prim), dd = wave_sum_u64(cnt.refl);
        if (lane == 0) {
            atomicAdd(&p.counters[0], (unsigned long long) a);
            atomicAdd(&p.counters[1], (unsigned long long) b);
            atomicAdd(&p.counters[2], (unsigned long long) c);
            atomicAdd(&p.counters[3], (unsigned long long) dd);
            const unsigned long long tEnd = __builtin_amdgcn_s_memrealtime();
            atomicAdd(&p.counters[14], tEnd - tStart);                         // sum of wave lifetimes
            atomicMax(&p.counters[15], tEnd);                                  // last wave to finish
            atomicMax(&p.counters[7], 0x4000000000000000ull - tStart);         // (2^62 - earliest start)
        }
    }
    if ((COUNT || RTD_CLK) && lane == 0) {
        atomicAdd(&p.counters[8], (unsigned long long) ss.refill);
        atomicAdd(&p.counters[9], (unsigned long long) ss.trips);
        atomicAdd(&p.counters[10], (unsigned long long) ss.leaf);
        atomicAdd(&p.counters[11], (unsigned long long) ss.shade);
        atomicAdd(&p.counters[12], (unsigned long long) ss.refillLanes);
        atomicAdd(&p.counters[13], (unsigned long long) ss.shadeLanes);
        atomicAdd(&p.counters[20], (unsigned long long) ss.slow);
        atomicAdd(&p.counters[21], (unsigned long long) ss.slowLanes);
        atomicAdd(&p.counters[22], (unsigned long long) ss.parkedLanes);
        atomicAdd(&p.counters[23], ss.tRefill);
        atomicAdd(&p.counters[24], ss.tSlow);
        atomicAdd(&p.counters[25], ss.tWalk);
        atomicAdd(&p.counters[26], ss.tShade);
        if (RTD_CLK) {
            atomicAdd(&p.counters[27], ss.tLoop);
            atomicAdd(&p.counters[28], ss.tLeaf);
            atomicAdd(&p.counters[29], ss.tUnb);
            atomicAdd(&p.counters[30], ss.tCam);
            atomicAdd(&p.counters[31], ss.tLamb);
            if (!COUNT) { atomicAdd(&p.counters[14], ss.loopTrips); atomicAdd(&p.counters[15], ss.loopLanes); } // (the counting variant keeps its wave lifetimes there)
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
