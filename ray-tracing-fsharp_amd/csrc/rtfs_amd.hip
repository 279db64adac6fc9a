// rtfs_amd.hip -- C ABI entry points of librtfs_amd.so (declared in include/rtfs_amd.h).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared (see Makefile).
// There is NO CPU fallback: every compute entry point returns RT_ERR_NO_DEVICE when no HIP device is visible.
#include "../../include/rtfs_amd.h"
#include "rt_device.h"
#include "rt_render_kernel.h"
#include "rt_scene.h"

#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <atomic>
#include <chrono>
#include <cstddef>
#include <cstdio>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

using namespace rtd;

// ------------------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }
#define HIP_TRY(expr)                                                                                                   \
    do {                                                                                                                \
        hipError_t e_ = (expr);                                                                                         \
        if (e_ != hipSuccess) return fail(RT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));               \
    } while (0)

static int visible_devices() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void) hipGetLastError(); return 0; }
    return n;
}
// Makes `device` current for the lifetime of the guard and puts the caller's device back afterwards (a torch caller that
// renders on another device than its current one must not find its later work redirected).
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    int enter(int device) {
        const int n = visible_devices();
        if (n <= 0) return fail(RT_ERR_NO_DEVICE, "no HIP device visible: the render path has no CPU fallback");
        if (device < 0 || device >= n) return fail(RT_ERR_INVALID_ARGUMENT, "device index out of range");
        if (hipGetDevice(&prev) != hipSuccess) { (void) hipGetLastError(); prev = -1; }
        if (prev != device) {
            hipError_t e = hipSetDevice(device);
            if (e != hipSuccess) return fail(RT_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
            switched = true;
        }
        return RT_OK;
    }
    ~DeviceGuard() { if (switched && prev >= 0) (void) hipSetDevice(prev); }
};

// ------------------------------------------------------------------------------------------------------------
// scene handle: host image + lazily created per-device copies
// ------------------------------------------------------------------------------------------------------------
struct DeviceScene {
    unsigned char *image = nullptr;
    TexRec *tex = nullptr;
    uint8_t *texels = nullptr;
    int cu_count = 0;
};
// Per-launch scratch, stream-ordered (hipMallocAsync on the launch stream): counters[16] | queues | camera.  Nothing is
// shared between launches, so any number of them may be in flight on any streams.
#define RT_SCRATCH_BYTES 512

struct rt_scene {
    rth::HostScene host;
    std::map<int, DeviceScene> dev;
    std::mutex mu;
};

// Process-wide DEFAULTS of the launch settings (rt_set_*): read once per call, and only for the rt_render_options fields a
// caller leaves at 0.
static std::atomic<int> g_block_threads{0}, g_chunk_pixels{0}, g_blocks_per_cu{0}, g_yield_lanes{0}, g_refill_lanes{0};
static std::atomic<int> g_passes{0}; // 0 auto, 1 fused kernel, 2 two-pass (A, sort, B)
static std::atomic<int> g_park_lanes{0};
static std::atomic<int> g_walk_tree{RT_WALK_TREE_SAH};
static thread_local unsigned long long g_last_stage_stats[16] = {0};

static int device_scene(rt_scene *s, int device, DeviceScene **out) {
    std::lock_guard<std::mutex> lock(s->mu);
    auto it = s->dev.find(device);
    if (it != s->dev.end()) { *out = &it->second; return RT_OK; }
    DeviceScene d;
    const rth::HostScene &h = s->host;
    HIP_TRY(hipMalloc((void **) &d.image, h.image.size()));
    HIP_TRY(hipMemcpy(d.image, h.image.data(), h.image.size(), hipMemcpyHostToDevice));
    if (!h.texRecs.empty()) {
        HIP_TRY(hipMalloc((void **) &d.tex, h.texRecs.size() * sizeof(TexRec)));
        HIP_TRY(hipMemcpy(d.tex, h.texRecs.data(), h.texRecs.size() * sizeof(TexRec), hipMemcpyHostToDevice));
    }
    if (!h.texelBlob.empty()) {
        HIP_TRY(hipMalloc((void **) &d.texels, h.texelBlob.size()));
        HIP_TRY(hipMemcpy(d.texels, h.texelBlob.data(), h.texelBlob.size(), hipMemcpyHostToDevice));
    }
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    d.cu_count = prop.multiProcessorCount;
    auto ins = s->dev.emplace(device, d);
    *out = &ins.first->second;
    return RT_OK;
}

// ------------------------------------------------------------------------------------------------------------
// launch plan: ONE place decides block size, unit sizes, LDS residency and the number of passes -- the launch uses it and
// rt_scene_get_info reports from it, so the two cannot disagree
// ------------------------------------------------------------------------------------------------------------
// LDS budget: 160 KiB per CU (MI355X_MICROARCH.md); the LDS part of the scene image plus every wave's scratch must fit one workgroup.
#define RT_LDS_BYTES 163840u
struct Settings { int block, chunk, blocks_per_cu, yield, refill, passes, park; };
static Settings resolve_settings(const rt_render_options *o) {
    rt_render_options v{};
    if (o) memcpy(&v, o, o->struct_size < sizeof(v) ? o->struct_size : sizeof(v));
    Settings s;
    s.block = v.block_threads ? v.block_threads : g_block_threads.load();
    s.chunk = v.chunk_pixels ? v.chunk_pixels : g_chunk_pixels.load();
    s.blocks_per_cu = v.blocks_per_cu ? v.blocks_per_cu : g_blocks_per_cu.load();
    s.yield = v.yield_lanes ? v.yield_lanes : g_yield_lanes.load();
    s.refill = v.refill_lanes ? v.refill_lanes : g_refill_lanes.load();
    s.passes = v.passes ? v.passes : g_passes.load();
    s.park = v.park_lanes ? v.park_lanes : g_park_lanes.load();
    return s;
}
static const char *check_settings(const Settings &s) {
    if (s.block != 0 && s.block != 256 && s.block != 512 && s.block != 768 && s.block != 1024) return "block_threads must be 0, 256, 512, 768 or 1024";
    if (s.chunk < 0 || s.chunk > RTD_MAX_CHUNK) return "chunk_pixels must be in [0, 64]";
    if (s.blocks_per_cu < 0 || s.blocks_per_cu > 8) return "blocks_per_cu must be in [0, 8]";
    if (s.yield < 0 || s.yield > 64 || s.refill < 0 || s.refill > 64) return "thresholds must be in [0, 64]";
    if (s.passes < 0 || s.passes > 2) return "passes must be 0 (auto), 1 (fused) or 2 (two-pass)";
    if (s.park < -1 || s.park > RTD_MAX_PARK) return "park_lanes must be in [-1, 256]";
    return nullptr;
}
// `count`: the counting kernel variant stages the exact double-precision node records (112 B), the timed one the single-precision
// filter records (64 B)
static size_t lds_need(const rth::HostScene &h, bool lds, bool count, int block, int chunk, bool passA = false) {
    return (lds ? (size_t) (count ? h.off.lds_total : h.off.lds32_total) : 0u) + (size_t) (block / 64) * (passA ? RTD_WAVE_WORDS_A(chunk) : RTD_WAVE_WORDS(chunk)) * 4u;
}
// The Lambert pool in LDS: as many 56-byte entries per wave as fit beside the scene and the waves' scratch, at most 64; with room
// for fewer than 32 the pool stays in global memory (entries of RTD_PARK_ENTRY_BYTES, L2-resident at best).  Returns the capacity
// and adds the pools' bytes to ldsBytes.
static int lambert_pool_lds(size_t &ldsBytes, int block) {
#ifdef RTD_NO_LDS_POOL
    return 0;
#endif
    if (ldsBytes >= RT_LDS_BYTES) return 0;
    const size_t waves = (size_t) block / 64u;
    size_t c = (RT_LDS_BYTES - ldsBytes) / (waves * RTD_PARK_L_LDS_BYTES);
    if (c > 64) c = 64;
    c &= ~(size_t) 1; // an even capacity keeps every field array 16-byte aligned
    if (c < 32) return 0;
    ldsBytes += waves * RTD_PARK_L_LDS_BYTES * c;
    return (int) c;
}
// The timed variant of a scene that is NOT LDS-resident keeps the first records of its depth-ordered node32 section in LDS
// (stage_nodes32, node_loop_glb32): what fits beside the waves' scratch (`ldsBytes` on entry) and a full Lambert pool.  Returns
// the bytes (a multiple of the record size; 0 for the counting variant, whose walk reads the exact records) and adds them.
static uint32_t hybrid_node_bytes(const rth::HostScene &h, size_t &ldsBytes, bool lds, bool count, int block, bool pool) {
#ifdef RTD_NO_HYBRID
    return 0u;
#endif
    if (lds || count) return 0u;
    const size_t poolBytes = pool ? (size_t) (block / 64) * RTD_PARK_L_LDS_BYTES * 64u : 0u; // (a pool of 48 measured the same, of 32 2 % slower)
    if (ldsBytes + poolBytes + RTD_NODE32_BYTES > RT_LDS_BYTES) return 0u;
    size_t room = (RT_LDS_BYTES - ldsBytes - poolBytes) & ~(size_t) (RTD_NODE32_BYTES - 1);
    const size_t all = (size_t) h.off.n_nodes * RTD_NODE32_BYTES;
    if (room > all) room = all;
    ldsBytes += room;
    return (uint32_t) room;
}
struct LaunchPlan {
    int block = 1024, chunk = 16, park = 0;
    bool lds = false;
};
// Block size and residency for a scene: LDS-resident if its image fits beside the waves' scratch at the preferred block, else the
// global-memory variant of the kernel at the same block (whose timed form keeps the top of the tree in LDS: hybrid_node_bytes).
// (Until round 3 a scene that fitted only beside the scratch of a 256-thread block was kept resident with such blocks: one wave
// per SIMD -- 899 spheres, tuned: 25.1 ms against 12.3 ms for the global-memory variant at 1024 threads.)
static LaunchPlan plan_launch(const rth::HostScene &h, const Settings &s, bool count = false) {
    LaunchPlan p;
    p.block = s.block ? s.block : 1024;
    p.chunk = s.chunk ? s.chunk : 16;
    p.park = s.park < 0 ? 0 : (s.park ? s.park : RTD_PARK_DEFAULT);
    // (an LDS-resident scene has far fewer than the 16384 objects the node loop's 14-bit queue entries can name: 48 B each of 160 KiB)
    auto fits = [&](int block, int chunk) { return h.nBounded + h.nUnbounded < 16384u && lds_need(h, true, count, block, chunk) <= RT_LDS_BYTES; };
    if (fits(p.block, p.chunk)) p.lds = true;
    return p; // (global-memory variant: the waves' scratch always fits)
}

// ------------------------------------------------------------------------------------------------------------
// launch
// ------------------------------------------------------------------------------------------------------------
typedef void (*render_fn)(const RenderParams);
template <int MODE, bool TEX> static render_fn pick_mode(bool lds, bool count, int block) {
    if (block == 1024) {
        if (lds) return count ? render_kernel<true, true, 1024, MODE, TEX> : render_kernel<true, false, 1024, MODE, TEX>;
        return count ? render_kernel<false, true, 1024, MODE, TEX> : render_kernel<false, false, 1024, MODE, TEX>;
    }
    if (block == 768) {
        if (lds) return count ? render_kernel<true, true, 768, MODE, TEX> : render_kernel<true, false, 768, MODE, TEX>;
        return count ? render_kernel<false, true, 768, MODE, TEX> : render_kernel<false, false, 768, MODE, TEX>;
    }
    if (block == 512) {
        if (lds) return count ? render_kernel<true, true, 512, MODE, TEX> : render_kernel<true, false, 512, MODE, TEX>;
        return count ? render_kernel<false, true, 512, MODE, TEX> : render_kernel<false, false, 512, MODE, TEX>;
    }
    if (lds) return count ? render_kernel<true, true, 256, MODE, TEX> : render_kernel<true, false, 256, MODE, TEX>;
    return count ? render_kernel<false, true, 256, MODE, TEX> : render_kernel<false, false, 256, MODE, TEX>;
}
// tex: the scene has parameterised textures (otherwise the variant compiled without the texture call: no scratch, no VGPR spills)
static render_fn pick_kernel(bool lds, bool count, int block, int mode, bool tex) {
    if (tex) return mode == 0 ? pick_mode<0, true>(lds, count, block) : (mode == 1 ? pick_mode<1, true>(lds, count, block) : (mode == 2 ? pick_mode<2, true>(lds, count, block) : pick_mode<3, true>(lds, count, block)));
    return mode == 0 ? pick_mode<0, false>(lds, count, block) : (mode == 1 ? pick_mode<1, false>(lds, count, block) : (mode == 2 ? pick_mode<2, false>(lds, count, block) : pick_mode<3, false>(lds, count, block)));
}

// Zeroes a launch's counters and queues and writes its camera: one tiny launch instead of a memset plus a copy from
// pageable host memory (which may block the host until earlier work on the stream has finished).
__global__ void launch_init_kernel(unsigned char *scratch, const CameraParams cam) {
    if (threadIdx.x < 64) ((unsigned int *) scratch)[threadIdx.x] = 0u; // 256 B: counters[16], queue, queue_b, live_count
    if (threadIdx.x == 0) *(CameraParams *) (scratch + 256) = cam;
}
static_assert(sizeof(CameraParams) <= RT_SCRATCH_BYTES - 256, "camera must fit the scratch slot");

extern "C" {

int rt_abi_version(void) { return RT_ABI_VERSION; }
const char *rt_last_error(void) { return g_err.c_str(); }
int rt_device_count(void) { return visible_devices(); }

size_t rt_abi_sizeof(int which) {
    switch (which) {
    case 0: return sizeof(rt_hittable);
    case 1: return sizeof(rt_texture);
    case 2: return sizeof(rt_camera);
    case 3: return sizeof(rt_scene_info);
    case 4: return sizeof(rt_stats);
    case 5: return sizeof(rt_render_options);
    case 6: return sizeof(rt_scene_options);
    case 7: return sizeof(rt_tune_info);
    default: return 0;
    }
}

size_t rt_abi_offsetof(int which, int field) {
#define RT_OFFS(T, ...) { static const size_t o[] = {__VA_ARGS__}; return (field >= 0 && (size_t) field < sizeof(o) / sizeof(o[0])) ? o[field] : (size_t) -1; }
#define F(T, m) offsetof(T, m)
    switch (which) {
    case 0: RT_OFFS(rt_hittable, F(rt_hittable, kind), F(rt_hittable, style), F(rt_hittable, point), F(rt_hittable, normal), F(rt_hittable, radius),
                    F(rt_hittable, albedo), F(rt_hittable, fuzz), F(rt_hittable, ior), F(rt_hittable, prob), F(rt_hittable, rgb), F(rt_hittable, reserved),
                    F(rt_hittable, texture))
    case 1: RT_OFFS(rt_texture, F(rt_texture, kind), F(rt_texture, rgb), F(rt_texture, ramp_src), F(rt_texture, reserved), F(rt_texture, even),
                    F(rt_texture, odd), F(rt_texture, grid_size), F(rt_texture, width), F(rt_texture, height), F(rt_texture, texels),
                    F(rt_texture, map_centre), F(rt_texture, map_radius))
    case 2: RT_OFFS(rt_camera, F(rt_camera, view_origin), F(rt_camera, view_dir), F(rt_camera, xaxis_origin), F(rt_camera, xaxis_dir),
                    F(rt_camera, yaxis_origin), F(rt_camera, yaxis_dir), F(rt_camera, viewport_width), F(rt_camera, viewport_height),
                    F(rt_camera, focal_length), F(rt_camera, samples_per_pixel), F(rt_camera, bounce_depth))
    case 3: RT_OFFS(rt_scene_info, F(rt_scene_info, n_bounded), F(rt_scene_info, n_unbounded), F(rt_scene_info, n_nodes), F(rt_scene_info, tree_depth),
                    F(rt_scene_info, n_textures), F(rt_scene_info, lds_resident), F(rt_scene_info, walk_tree), F(rt_scene_info, walk_tree_depth),
                    F(rt_scene_info, scene_bytes), F(rt_scene_info, texel_bytes), F(rt_scene_info, walk_tree_nodes), F(rt_scene_info, leaf_box_implied))
    case 4: RT_OFFS(rt_stats, F(rt_stats, rays), F(rt_stats, aabb_tests), F(rt_stats, prim_tests), F(rt_stats, reflections), F(rt_stats, samples),
                    F(rt_stats, pixels), F(rt_stats, pixels_early), F(rt_stats, kernel_ms), F(rt_stats, total_ms))
    case 5: RT_OFFS(rt_render_options, F(rt_render_options, struct_size), F(rt_render_options, block_threads), F(rt_render_options, chunk_pixels),
                    F(rt_render_options, blocks_per_cu), F(rt_render_options, yield_lanes), F(rt_render_options, refill_lanes),
                    F(rt_render_options, passes), F(rt_render_options, park_lanes))
    case 6: RT_OFFS(rt_scene_options, F(rt_scene_options, struct_size), F(rt_scene_options, walk_tree))
    case 7: RT_OFFS(rt_tune_info, F(rt_tune_info, struct_size), F(rt_tune_info, tuned), F(rt_tune_info, probe_rows), F(rt_tune_info, probe_rays),
                    F(rt_tune_info, nodes_before), F(rt_tune_info, nodes_after), F(rt_tune_info, box_tests_before), F(rt_tune_info, box_tests_after),
                    F(rt_tune_info, probe_ms), F(rt_tune_info, build_ms))
    default: return (size_t) -1;
    }
#undef F
#undef RT_OFFS
}

int rt_set_launch_config(int32_t block_threads, int32_t chunk_pixels, int32_t blocks_per_cu) {
    Settings s{block_threads, chunk_pixels, blocks_per_cu, 0, 0, 0, 0};
    if (const char *m = check_settings(s)) return fail(RT_ERR_INVALID_ARGUMENT, m);
    g_block_threads = block_threads;
    g_chunk_pixels = chunk_pixels;
    g_blocks_per_cu = blocks_per_cu;
    return RT_OK;
}

int rt_last_stage_stats(uint64_t out[16]) {
    if (!out) return fail(RT_ERR_INVALID_ARGUMENT, "NULL argument");
    for (int i = 0; i < 16; ++i) out[i] = g_last_stage_stats[i];
    return RT_OK;
}

int rt_set_walk_tree(int32_t kind) {
    if (kind != RT_WALK_TREE_SAH && kind != RT_WALK_TREE_REFERENCE) return fail(RT_ERR_INVALID_ARGUMENT, "walk tree must be RT_WALK_TREE_SAH or RT_WALK_TREE_REFERENCE");
    g_walk_tree = kind;
    return RT_OK;
}
int rt_set_passes(int32_t passes) {
    if (passes < 0 || passes > 2) return fail(RT_ERR_INVALID_ARGUMENT, "passes must be 0 (auto), 1 (fused) or 2 (two-pass)");
    g_passes = passes;
    return RT_OK;
}
int rt_set_park(int32_t park_lanes) {
    if (park_lanes < -1 || park_lanes > RTD_MAX_PARK) return fail(RT_ERR_INVALID_ARGUMENT, "park_lanes must be in [-1, 256]");
    g_park_lanes = park_lanes;
    return RT_OK;
}

int rt_set_schedule(int32_t yield_lanes, int32_t refill_lanes) {
    if (yield_lanes < 0 || yield_lanes > 64 || refill_lanes < 0 || refill_lanes > 64) return fail(RT_ERR_INVALID_ARGUMENT, "thresholds must be in [0, 64]");
    g_yield_lanes = yield_lanes;
    g_refill_lanes = refill_lanes;
    return RT_OK;
}

int rt_camera_make_basic(int32_t spp, double focal, double aspect, const double origin[3], const double view_direction[3],
                         const double view_up[3], rt_camera *out) {
    if (!origin || !view_direction || !view_up || !out) return fail(RT_ERR_INVALID_ARGUMENT, "NULL argument");
    if (!rth::camera_make_basic(spp, focal, aspect, origin, view_direction, view_up, out))
        return fail(RT_ERR_INVALID_ARGUMENT, "degenerate camera frame (the reference's ValueOption.get would throw)");
    return RT_OK;
}

int rt_scene_create_ex(const rt_hittable *hittables, size_t n_hittables, const rt_texture *textures, size_t n_textures,
                       const rt_scene_options *options, rt_scene **out) {
    if (!out) return fail(RT_ERR_INVALID_ARGUMENT, "out is NULL");
    int walk = -1;
    if (options && options->struct_size >= offsetof(rt_scene_options, walk_tree) + sizeof(int32_t)) walk = options->walk_tree;
    if (walk == -1) walk = g_walk_tree.load();
    if (walk != RT_WALK_TREE_SAH && walk != RT_WALK_TREE_REFERENCE) return fail(RT_ERR_INVALID_ARGUMENT, "walk_tree must be RT_WALK_TREE_SAH, RT_WALK_TREE_REFERENCE or -1");
    std::unique_ptr<rt_scene> s(new rt_scene());
    int status = RT_OK;
    std::string msg = rth::build_scene(hittables, n_hittables, textures, n_textures, walk, s->host, status);
    if (status != RT_OK) return fail(status, msg);
    *out = s.release();
    return RT_OK;
}
int rt_scene_create(const rt_hittable *hittables, size_t n_hittables, const rt_texture *textures, size_t n_textures, rt_scene **out) {
    return rt_scene_create_ex(hittables, n_hittables, textures, n_textures, nullptr, out);
}

void rt_scene_destroy(rt_scene *s) {
    if (!s) return;
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) { (void) hipGetLastError(); prev = -1; }
    for (auto &kv : s->dev) {
        if (hipSetDevice(kv.first) != hipSuccess) continue;
        (void) hipFree(kv.second.image);
        (void) hipFree(kv.second.tex);
        (void) hipFree(kv.second.texels);
    }
    if (prev >= 0 && !s->dev.empty()) (void) hipSetDevice(prev);
    delete s;
}

int rt_scene_get_info(const rt_scene *s, rt_scene_info *out) {
    if (!s || !out) return fail(RT_ERR_INVALID_ARGUMENT, "NULL argument");
    const rth::HostScene &h = s->host;
    out->n_bounded = h.off.n_bounded;
    out->n_unbounded = h.off.n_unbounded;
    out->n_nodes = (int32_t) h.tree.skip.size();
    out->walk_tree_nodes = h.off.n_nodes;
    out->leaf_box_implied = h.off.box_implied;
    out->tree_depth = h.tree.depth;
    out->walk_tree = h.walkKind;
    out->walk_tree_depth = h.walkTree.depth;
    out->n_textures = (int32_t) h.texRecs.size();
    out->lds_resident = plan_launch(h, resolve_settings(nullptr)).lds ? 1 : 0; // the decision a render with default options takes
    out->scene_bytes = (int64_t) h.off.total;
    out->texel_bytes = (int64_t) h.texelBlob.size();
    return RT_OK;
}

static int copy_tree(const rth::HostScene &h, const rth::FlatTree &t, int32_t *skip, int32_t *prim, double *boxes) {
    const size_t nn = t.skip.size();
    for (size_t i = 0; i < nn; ++i) {
        if (skip) skip[i] = t.skip[i];
        if (prim) prim[i] = t.prim[i] < 0 ? -1 : h.objToOrig[(size_t) t.prim[i]];
        if (boxes) for (int a = 0; a < 3; ++a) { boxes[i * 6 + (size_t) a * 2] = t.box[i].mn[a]; boxes[i * 6 + (size_t) a * 2 + 1] = t.box[i].mx[a]; }
    }
    return RT_OK;
}
int rt_scene_get_tree(const rt_scene *s, int32_t *skip, int32_t *prim, double *boxes) {
    if (!s) return fail(RT_ERR_INVALID_ARGUMENT, "NULL scene");
    return copy_tree(s->host, s->host.tree, skip, prim, boxes);
}
int rt_scene_get_walk_tree(const rt_scene *s, int32_t *skip, int32_t *prim, double *boxes) {
    if (!s) return fail(RT_ERR_INVALID_ARGUMENT, "NULL scene");
    return copy_tree(s->host, s->host.walkTree, skip, prim, boxes);
}

int rt_scene_get_filter_tree(const rt_scene *s, float *boxes, int32_t *links) {
    if (!s) return fail(RT_ERR_INVALID_ARGUMENT, "NULL scene");
    if (!boxes || !links) return fail(RT_ERR_INVALID_ARGUMENT, "NULL output");
    const rth::HostScene &h = s->host;
    const unsigned char *sec = h.image.data() + h.off.node32;
    for (int32_t p = 0; p < h.off.n_nodes; ++p) {
        const float *fx = (const float *) (sec + (size_t) p * RTD_NODE32_BYTES);
        const int32_t *lk = (const int32_t *) (sec + (size_t) p * RTD_NODE32_BYTES + 48);
        for (int a = 0; a < 3; ++a) { boxes[(size_t) p * 6 + 2 * a] = fx[a * 4]; boxes[(size_t) p * 6 + 2 * a + 1] = fx[a * 4 + 1]; }
        const uint32_t e = (uint32_t) lk[2];
        const uint32_t obj = (e & RTD_PEND_WIDE) ? (e & ~RTD_PEND_WIDE) : (e & (RTD_PEND_MARK - 1u));
        links[(size_t) p * 5 + 0] = lk[0] / RTD_NODE32_BYTES;
        links[(size_t) p * 5 + 1] = lk[1] / RTD_NODE32_BYTES;
        links[(size_t) p * 5 + 2] = lk[2];
        links[(size_t) p * 5 + 3] = lk[3];
        links[(size_t) p * 5 + 4] = e == 0u ? -1 : (obj < h.objToOrig.size() ? h.objToOrig[obj] : -2);
    }
    return RT_OK;
}

static int check_geometry(const rt_camera *camera, int32_t max_w, int32_t max_h, int32_t row_first, int32_t row_stride, int32_t n_rows) {
    if (!camera) return fail(RT_ERR_INVALID_ARGUMENT, "camera is NULL");
    if (max_w <= 0 || max_h <= 0) return fail(RT_ERR_INVALID_ARGUMENT, "max_width_coord and max_height_coord must be positive");
    if (max_w > (1 << 20) || max_h > (1 << 20)) return fail(RT_ERR_INVALID_ARGUMENT, "image too large");
    if (camera->samples_per_pixel < 1) return fail(RT_ERR_INVALID_ARGUMENT, "samples_per_pixel must be >= 1");
    if (camera->samples_per_pixel > 8000000) return fail(RT_ERR_INVALID_ARGUMENT, "samples_per_pixel too large for int32 sums (255*spp)");
    if (camera->bounce_depth < 0) return fail(RT_ERR_INVALID_ARGUMENT, "bounce_depth must be >= 0");
    if (camera->bounce_depth > 0xFFFFFF) return fail(RT_ERR_INVALID_ARGUMENT, "bounce_depth too large");
    const int rows = 2 * max_h + 1;
    if (row_stride <= 0 || row_first < 0 || n_rows < 0) return fail(RT_ERR_INVALID_ARGUMENT, "bad row shard");
    if (n_rows > 0 && (int64_t) row_first + (int64_t) (n_rows - 1) * row_stride >= rows) return fail(RT_ERR_INVALID_ARGUMENT, "row shard exceeds the image");
    return RT_OK;
}

} // extern "C"

// What a launch leaves behind when its statistics are wanted: events around the kernels and the launch's scratch, which
// then stays allocated until the counters have been read.
struct RayLog { double *rays; unsigned int *count; uint32_t cap, mask; }; // rt_scene_tune's probe (RenderParams::ray_log)

struct Pending {
    hipEvent_t a = nullptr, b = nullptr;
    unsigned char *scr = nullptr;
    hipStream_t st = nullptr;
    int device = -1;
    uint64_t pixels = 0, waves = 0;
    bool launched = false, keep = false;
    std::chrono::steady_clock::time_point t0;
    void release() { // events destroyed, scratch handed back to the stream's pool in stream order
        if (a) (void) hipEventDestroy(a);
        if (b) (void) hipEventDestroy(b);
        if (scr) (void) hipFreeAsync(scr, st);
        a = b = nullptr; scr = nullptr;
    }
    ~Pending() { release(); }
};

// Enqueues one shard's render on `stream`; never waits for the device.  With want_stats the launch is bracketed by events
// and its scratch is kept in `pd` for collect_stats.
static int launch_render(const rt_scene *scene, const rt_camera *camera, int32_t max_w, int32_t max_h, uint64_t seed, int32_t device,
                         int32_t row_first, int32_t row_stride, int32_t n_rows, uint32_t flags, void *d_accum, void *d_rgb, void *stream,
                         const rt_render_options *options, bool want_stats, Pending &pd, const RayLog *log = nullptr) {
    rt_stats *stats = want_stats ? (rt_stats *) 1 : nullptr; // only tested for NULL below
    if (!scene) return fail(RT_ERR_INVALID_ARGUMENT, "scene is NULL");
    int rc = check_geometry(camera, max_w, max_h, row_first, row_stride, n_rows);
    if (rc != RT_OK) return rc;
    if (n_rows > 0 && !d_accum) return fail(RT_ERR_INVALID_ARGUMENT, "d_accum is NULL");
    if (options && options->struct_size < sizeof(uint32_t)) return fail(RT_ERR_INVALID_ARGUMENT, "rt_render_options.struct_size is not set");
    const Settings set = resolve_settings(options);
    if (const char *m = check_settings(set)) return fail(RT_ERR_INVALID_ARGUMENT, m);
    DeviceGuard guard;
    rc = guard.enter(device);
    if (rc != RT_OK) return rc;
    const auto t0 = std::chrono::steady_clock::now();
    DeviceScene *ds = nullptr;
    rc = device_scene(const_cast<rt_scene *>(scene), device, &ds);
    if (rc != RT_OK) return rc;
    const rth::HostScene &h = scene->host;
    hipStream_t st = (hipStream_t) stream;

    const bool count = (flags & RT_RENDER_COUNTERS) != 0;
    const LaunchPlan plan = plan_launch(h, set, count);
    const int block = plan.block;
    const bool lds = plan.lds;
    // Units of the fused kernel: 16 pixels -- except for frames of few samples per pixel (always fused: the two-pass rule below needs
    // at least 64 samples in phase 2).  A unit is drained before the next, and one of 16 pixels x 11 samples is three rounds of a wave that
    // then waits for its longest path: as wide as still leaves a wave seven units, the scene in LDS permitting (2401x1601 px:
    // 4 spp 8.5 -> 16.0 Gray/s, 16 spp 12.3 -> 19.6 with 64 pixels; 1201x801: 7.0 -> 9.3, 10.5 -> 12.5 with 32).
    int chunk = plan.chunk;
    if (!set.chunk && set.passes != 2 && camera->samples_per_pixel <= 74) { // (below the two-pass rule's 64 samples in phase 2)
        const uint64_t px = (uint64_t) n_rows * (uint64_t) (2 * max_w + 1), waves = (uint64_t) ds->cu_count * (uint64_t) (block / 64);
        for (int c = 64; c > chunk; c /= 2)
            if (px >= 7ull * (uint64_t) c * waves && (!lds || lds_need(h, true, count, block, c) <= RT_LDS_BYTES)) { chunk = c; break; }
    }

    RenderParams p{};
    CameraParams hostCam{};
    for (int a = 0; a < 3; ++a) {
        hostCam.eye[a] = camera->view_origin[a];
        hostCam.xo[a] = camera->xaxis_origin[a];
        hostCam.xd[a] = camera->xaxis_dir[a];
        hostCam.yd[a] = camera->yaxis_dir[a];
    }
    hostCam.vw = camera->viewport_width;
    hostCam.vh = camera->viewport_height;
    hostCam.max_w = max_w; hostCam.max_h = max_h;
    hostCam.spp = camera->samples_per_pixel;
    hostCam.depth = camera->bounce_depth;
    p.max_w = max_w; p.max_h = max_h;
    p.spp = camera->samples_per_pixel;
    p.depth = camera->bounce_depth;
    p.off = h.off;
    p.scene_image = ds->image;
    p.tex = ds->tex;
    p.texels = ds->texels;
    p.seed_key = mix64(seed + 0x9E3779B97F4A7C15ull); // seed_key(), host side
    p.cols = 2 * max_w + 1;
    p.row_first = row_first; p.row_stride = row_stride; p.n_rows = n_rows;
    const int half = camera->samples_per_pixel / 2;
    p.k = half < 5 ? half : 5; // min 5 (spp / 2), Scene.fs:172
    p.chunk = chunk;
    p.park = plan.park;
    p.park_l = plan.park > 0 ? RTD_PARK_L_DEFAULT : 0; // the Lambert pool rides with the general one ("never park" switches both off)
    p.yield_lanes = set.yield ? set.yield : RTD_YIELD_DEFAULT;
    // the node loop runs on a little past the point where the STAGE would yield before it hands over to a leaf pass: fewer, fuller
    // leaf passes (measured on the bench frame, yield / hand-over: 52/52 110.9 ms, 52/56 109.9, 50/55 109.5, 48/56 109.5, 50/58 110.4)
    p.leaf_wait = p.yield_lanes + RTD_LEAF_WAIT_EXTRA > 64 ? 64 : p.yield_lanes + RTD_LEAF_WAIT_EXTRA;
    p.refill_lanes = set.refill ? set.refill : RTD_REFILL_DEFAULT;
    p.accum = (int32_t *) d_accum;
    p.rgb = (uint8_t *) d_rgb;
    if (log) { p.ray_log = log->rays; p.ray_log_count = log->count; p.ray_log_cap = log->cap; p.ray_log_mask = log->mask; }

    const bool tex = !h.texRecs.empty();
    render_fn fn = pick_kernel(lds, count, block, log ? 3 : 0, tex); // (the ray log of rt_scene_tune's probe: a kernel of its own)
    size_t ldsBytes = lds_need(h, lds, count, block, chunk);
    p.lds_node_bytes = (int32_t) hybrid_node_bytes(h, ldsBytes, lds, count, block, p.park_l > 0);
    p.lds_node_thr = RTD_HYBRID_LANES;
    if (p.park_l > 0) { // the fused launch's Lambert pool: in LDS if it fits (the two-pass launches decide for themselves below)
        const int cl = lambert_pool_lds(ldsBytes, block);
        if (cl) { p.park_l = cl; p.park_l_lds = 1; }
    }
    HIP_TRY(hipFuncSetAttribute((const void *) fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int) ldsBytes));
    int perCu = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, (const void *) fn, block, ldsBytes));
    if (perCu < 1) return fail(RT_ERR_HIP, "render kernel does not fit on a CU (occupancy 0)");
    if (set.blocks_per_cu > 0 && set.blocks_per_cu < perCu) perCu = set.blocks_per_cu;
    const uint64_t nLocal = (uint64_t) n_rows * (uint64_t) p.cols;
    const uint64_t units = (nLocal + (uint64_t) chunk - 1) / (uint64_t) chunk;
    const uint64_t fullGrid = (uint64_t) ds->cu_count * (uint64_t) perCu;
    uint64_t grid = fullGrid;
    const uint64_t wavesPerBlock = (uint64_t) block / 64u;
    const uint64_t needBlocks = (units + wavesPerBlock - 1) / wavesPerBlock;
    if (grid > needBlocks) grid = needBlocks;
    // Few units per wave => the fused kernel ends with most waves waiting for a few long units (a 16-pixel unit on a glass sphere
    // takes tens of ms): render in two passes with the second one ordered longest-job-first.  Many units per wave => the fused
    // kernel's tail is ~2 % and it saves the second launch (break-even measured at ~60 units per wave: config 3 whole, 59 per wave,
    // 221 ms in two passes against 227 ms fused; config 4 whole, 507 per wave, 3.75 s against 3.66 s).  spp <= 2k+1 has no second phase at all, and with few remaining samples
    // per pixel a unit is short, so is the tail, and the second launch costs more than it removes (config 2, 100 spp: fused 3.0 ms,
    // two passes 3.5 ms; config 3's 1/8 shard, 500 spp: 51 ms against 30 ms).
    const int n2 = camera->samples_per_pixel - 2 * p.k - 1;
    const bool twoPass = n2 > 0 && nLocal > 0 && nLocal < (1ull << 32) &&
                         (set.passes == 2 || (set.passes == 0 && (n2 >= 128 || (n2 >= 64 && nLocal >= (1ull << 21))) && units < 64ull * fullGrid * wavesPerBlock));
    // (frames of 2 Mpx and more pay for the second launch from ~75 spp: 2401x1601 at 100 spp 23.1 Gray/s fused, 26.9 in two passes; 1201x801: 19.6 / 19.8)

    // Everything below is stream-ordered: scratch and workspace come from the stream's pool and go back to it after the last
    // launch that uses them, so no launch shares state with another and the call returns without waiting for the device.
    const size_t pairsBytes = twoPass ? (((size_t) nLocal * 8u + 15u) & ~(size_t) 15u) : 0u, listBytes = twoPass ? (((size_t) nLocal * 4u + 15u) & ~(size_t) 15u) : 0u;
    const size_t sortBytes = twoPass ? (3u * RTD_COST_BUCKETS * 4u + 15u) & ~(size_t) 15u : 0u;
    const size_t poolBytes = (size_t) (twoPass ? fullGrid : grid) * (size_t) wavesPerBlock * (size_t) RTD_PARK_ENTRY_BYTES *
                             (size_t) (p.park + (p.park > 0 ? RTD_PARK_L_DEFAULT : 0) + (tex ? p.park : 0)); // general + Lambert (unless in LDS) + textured
    unsigned char *scr = nullptr;
    if (grid > 0 || stats) HIP_TRY(hipMallocAsync((void **) &scr, RT_SCRATCH_BYTES + pairsBytes + listBytes + sortBytes + poolBytes, st));
    Pending &cl = pd; // on every exit path its destructor (or collect_stats) gives the scratch back
    cl.scr = scr; cl.st = st; cl.device = device; cl.t0 = t0;
    cl.pixels = nLocal;
    cl.waves = (twoPass ? fullGrid : grid) * wavesPerBlock;
    if (stats) {
        HIP_TRY(hipEventCreate(&cl.a));
        HIP_TRY(hipEventCreate(&cl.b));
    }
    unsigned char *ws = scr ? scr + RT_SCRATCH_BYTES : nullptr;
    p.counters = (unsigned long long *) scr;
    p.queue = (unsigned int *) (scr + 128);
    p.cam_ptr = (const CameraParams *) (scr + 256);
    p.park_pool = ws ? ws + pairsBytes + listBytes + sortBytes : nullptr;
    if (scr) {
        hipLaunchKernelGGL(launch_init_kernel, dim3(1), dim3(64), 0, st, scr, hostCam);
        HIP_TRY(hipGetLastError());
    }
    if (grid > 0) {
        if (stats) HIP_TRY(hipEventRecord(cl.a, st));
        if (!twoPass) {
            hipLaunchKernelGGL(fn, dim3((unsigned) grid), dim3((unsigned) block), ldsBytes, st, p);
            HIP_TRY(hipGetLastError());
        } else {
            // workspace: pairs[nLocal] u64, list[nLocal] u32, hist/offsets/cursor[64] u32
            unsigned int *sortBuf = (unsigned int *) (ws + pairsBytes + listBytes);
            HIP_TRY(hipMemsetAsync(sortBuf, 0, sortBytes, st));
            p.pairs = (unsigned long long *) ws;
            p.live_list = (const unsigned int *) (ws + pairsBytes);
            p.queue_b = (unsigned long long *) (scr + 136);
            p.live_count = (unsigned int *) (scr + 144);
            p.total_waves = (uint32_t) (fullGrid * wavesPerBlock);
            render_fn fa = pick_kernel(lds, count, block, 1, tex), fb = pick_kernel(lds, count, block, 2, tex);
            // Unit sizes: pass A traces only 2k+1 samples per pixel, so its units are wide (below); pass B's largest unit is about a
            // sixteenth of a wave's share of the shard (measured best: 32 px at 1/2 frame, 16 at 1/4, 8 at 1/8 of config 3),
            // and shrinks towards the end of the cost-ordered list.
            int chunkA = set.chunk ? set.chunk : 64, chunkB = set.chunk ? set.chunk : 4;
            if (!set.chunk) {
                // pass A drains every unit before the next (its last paths run with most lanes idle), so wide units pay -- as long as a
                // wave still gets seven or so of them (measured: whole frame 64 px 5.4 ms, 32 px 6.2, 16 px 8.2; an eighth: 16 px best)
                while (chunkA > 8 && nLocal < 7ull * (uint64_t) chunkA * fullGrid * wavesPerBlock) chunkA /= 2;
                const uint64_t share = nLocal / (fullGrid * wavesPerBlock * 16u);
                while (chunkB < 32 && (uint64_t) chunkB * 3u / 2u <= share) chunkB *= 2; // nearest power of two
            }
            // both passes must fit the LDS beside the scene image, decided BEFORE anything is launched (a misfit found after
            // pass A would leave a half-rendered buffer); 
            while (lds && chunkA > 1 && lds_need(h, true, count, block, chunkA, true) > RT_LDS_BYTES) chunkA /= 2;
            while (lds && chunkB > 1 && lds_need(h, true, count, block, chunkB) > RT_LDS_BYTES) chunkB /= 2;
            if (lds && plan.park > 0 && !set.chunk) { // ... and not so wide that the Lambert pool no longer fits beside them
                auto pool_fits = [&](int c) { size_t b = lds_need(h, true, count, block, c, true); return lambert_pool_lds(b, block) != 0; };
                while (chunkA > 16 && !pool_fits(chunkA) && pool_fits(chunkA / 2)) chunkA /= 2;
            }
            size_t ldsA = lds_need(h, lds, count, block, chunkA, true), ldsB = lds_need(h, lds, count, block, chunkB);
            if (lds && (ldsA > RT_LDS_BYTES || ldsB > RT_LDS_BYTES)) return fail(RT_ERR_HIP, "two-pass launch does not fit the LDS");
            const uint32_t hybA = hybrid_node_bytes(h, ldsA, lds, count, block, plan.park > 0), hybB = hybrid_node_bytes(h, ldsB, lds, count, block, plan.park > 0);
            int clA = 0, clB = 0;
            if (plan.park > 0) { clA = lambert_pool_lds(ldsA, block); clB = lambert_pool_lds(ldsB, block); }
            HIP_TRY(hipFuncSetAttribute((const void *) fa, hipFuncAttributeMaxDynamicSharedMemorySize, (int) ldsA));
            HIP_TRY(hipFuncSetAttribute((const void *) fb, hipFuncAttributeMaxDynamicSharedMemorySize, (int) ldsB));
            RenderParams pa = p;
            pa.chunk = chunkA;
            pa.lds_node_bytes = (int32_t) hybA; p.lds_node_bytes = (int32_t) hybB;
            pa.park_l = clA ? clA : (plan.park > 0 ? RTD_PARK_L_DEFAULT : 0); pa.park_l_lds = clA ? 1 : 0;
            p.park_l = clB ? clB : (plan.park > 0 ? RTD_PARK_L_DEFAULT : 0); p.park_l_lds = clB ? 1 : 0;
            const uint64_t unitsA = (nLocal + (uint64_t) chunkA - 1) / (uint64_t) chunkA;
            uint64_t gridA = (unitsA + wavesPerBlock - 1) / wavesPerBlock;
            if (gridA > fullGrid) gridA = fullGrid;
            hipLaunchKernelGGL(fa, dim3((unsigned) gridA), dim3((unsigned) block), ldsA, st, pa);
            p.chunk = chunkB;
            hipLaunchKernelGGL(sort_hist_kernel, dim3(256), dim3(256), 0, st, (const unsigned long long *) p.pairs, (const unsigned int *) p.live_count,
                               (uint32_t) (2 * p.k + 1), sortBuf);
            hipLaunchKernelGGL(sort_offsets_kernel, dim3(1), dim3(64), 0, st, (const unsigned int *) sortBuf, sortBuf + RTD_COST_BUCKETS);
            hipLaunchKernelGGL(sort_scatter_kernel, dim3(256), dim3(256), 0, st, (const unsigned long long *) p.pairs, (const unsigned int *) p.live_count,
                               (uint32_t) (2 * p.k + 1), (const unsigned int *) (sortBuf + RTD_COST_BUCKETS), sortBuf + 2 * RTD_COST_BUCKETS,
                               (unsigned int *) (ws + pairsBytes));
            hipLaunchKernelGGL(fb, dim3((unsigned) fullGrid), dim3((unsigned) block), ldsB, st, p);
            const hipError_t e = hipGetLastError();
            if (e != hipSuccess) return fail(RT_ERR_HIP, std::string("two-pass launch: ") + hipGetErrorString(e));
        }
        if (stats) HIP_TRY(hipEventRecord(cl.b, st));
    }
    cl.launched = grid > 0;
    if (!stats) cl.release();
    return RT_OK;
}

// Waits for the launch's stream and reads its counters (the device of the launch must be current).
static int collect_stats(Pending &pd, rt_stats *stats) {
    HIP_TRY(hipStreamSynchronize(pd.st));
    unsigned long long c[32] = {0};
    if (pd.scr) {
        HIP_TRY(hipMemcpy(c, pd.scr, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(c + 16, pd.scr + 160, 12 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    }
#ifdef RTD_STAGE_CLOCKS
    if (getenv("RTFS_STAGE_CLOCKS")) // diagnostic build: the finer clocks of rt_render_kernel.h's StageStats
        fprintf(stderr, "stage clocks: loop %llu leaf %llu unbounded %llu new_items %llu lambert %llu; node loop trips %llu, lanes stepping %llu (%.1f of 64 per trip)\n",
                c[23], c[24], c[25], c[26], c[27], c[14], c[15], c[14] ? (double) c[15] / (double) c[14] : 0.0);
#endif
    for (int i = 0; i < 6; ++i) g_last_stage_stats[i] = c[8 + i];
    g_last_stage_stats[6] = c[14];                                  // sum of wave lifetimes, 100 MHz ticks
    g_last_stage_stats[7] = c[15] - (0x4000000000000000ull - c[7]); // first wave start -> last wave end, ticks
    g_last_stage_stats[8] = pd.waves;
    for (int i = 0; i < 7; ++i) g_last_stage_stats[9 + i] = c[16 + i]; // slow stages, lanes in them, lanes parked, cycles in refill / slow / walk / shade
    float ms = 0.f;
    if (pd.launched) HIP_TRY(hipEventElapsedTime(&ms, pd.a, pd.b));
    memset(stats, 0, sizeof(*stats));
    stats->rays = c[0]; stats->aabb_tests = c[1]; stats->prim_tests = c[2]; stats->reflections = c[3];
    stats->samples = c[4]; stats->pixels_early = c[5];
    stats->pixels = pd.pixels;
    stats->kernel_ms = ms;
    stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - pd.t0).count();
    pd.release();
    return RT_OK;
}

extern "C" {

int rt_render_device_ex(const rt_scene *scene, const rt_camera *camera, int32_t max_w, int32_t max_h, uint64_t seed, int32_t device,
                        int32_t row_first, int32_t row_stride, int32_t n_rows, uint32_t flags, void *d_accum, void *d_rgb, void *stream,
                        const rt_render_options *options, rt_stats *stats) {
    DeviceGuard guard; // entered again inside launch_render (a no-op then); kept here so that collect_stats runs on the device too
    int rc = guard.enter(device);
    if (rc != RT_OK) return rc;
    Pending pd;
    rc = launch_render(scene, camera, max_w, max_h, seed, device, row_first, row_stride, n_rows, flags, d_accum, d_rgb, stream, options, stats != nullptr, pd);
    if (rc != RT_OK || !stats) return rc;
    return collect_stats(pd, stats);
}

int rt_render_device(const rt_scene *scene, const rt_camera *camera, int32_t max_w, int32_t max_h, uint64_t seed, int32_t device,
                     int32_t row_first, int32_t row_stride, int32_t n_rows, uint32_t flags, void *d_accum, void *d_rgb, void *stream,
                     rt_stats *stats) {
    return rt_render_device_ex(scene, camera, max_w, max_h, seed, device, row_first, row_stride, n_rows, flags, d_accum, d_rgb, stream, nullptr, stats);
}

} // extern "C"

// The host half of tuning: at most 8192 of `raw` (evenly spaced) become the probe (the tree's quality levels off between 4096 and
// 8192 rays -- 15.64 / 15.50 box tests per held-out ray of the final scene, 15.55 with 32768 -- and the build time is linear in them); the walk tree is rebuilt and every device that
// holds a copy of the image gets the new one, once it has finished what it was doing.
struct RawRay { double v[6]; };
static int apply_tune(rt_scene *scene, const std::vector<RawRay> &raw, rt_tune_info &out) {
    const auto t0 = std::chrono::steady_clock::now();
    const size_t want = 8192;
    std::vector<rth::ProbeRay> rays;
    const size_t n = raw.size() < want ? raw.size() : want;
    rays.reserve(n);
    for (size_t i = 0; i < n; ++i) {
        const RawRay &r = raw[i * raw.size() / n];
        rth::ProbeRay pr;
        for (int a = 0; a < 3; ++a) { pr.o[a] = r.v[a]; pr.inv[a] = 1.0 / r.v[3 + a]; }
        rays.push_back(pr);
    }
    rth::TuneResult tr;
    std::lock_guard<std::mutex> lock(scene->mu);
    rth::HostScene &h = scene->host;
    // Failure-atomic: the host tree and every device copy change together or not at all.  The new image is uploaded to fresh
    // allocations first; only when every device has its copy are the old ones released and the pointers swapped.  On any failure
    // the host scene is put back as it was and the new allocations are freed, so no launch can pair an old image with new offsets.
    struct Saved { rth::FlatTree walkTree; int walkKind; std::vector<unsigned char> image; rtd::SceneOffsets off; } saved{h.walkTree, h.walkKind, h.image, h.off};
    if (!rth::tune_walk_tree(h, rays, tr)) return RT_OK; // a reference-tree scene, or no rays: left as it is
    if (!scene->dev.empty()) {
        int prev = -1;
        if (hipGetDevice(&prev) != hipSuccess) { (void) hipGetLastError(); prev = -1; }
        struct Restore { int prev; ~Restore() { if (prev >= 0) (void) hipSetDevice(prev); } } restore{prev};
        std::vector<std::pair<int, unsigned char *>> fresh;
        hipError_t err = hipSuccess;
        for (auto &kv : scene->dev) {
            unsigned char *img = nullptr;
            err = hipSetDevice(kv.first);
            if (err == hipSuccess) err = hipMalloc((void **) &img, h.image.size());
            if (err == hipSuccess) { fresh.emplace_back(kv.first, img); err = hipMemcpy(img, h.image.data(), h.image.size(), hipMemcpyHostToDevice); }
            if (err != hipSuccess) break;
        }
        if (err != hipSuccess) {
            for (auto &f : fresh) if (hipSetDevice(f.first) == hipSuccess) (void) hipFree(f.second);
            h.walkTree = std::move(saved.walkTree); h.walkKind = saved.walkKind; h.image = std::move(saved.image); h.off = saved.off;
            return fail(RT_ERR_HIP, std::string("rt_scene_tune: uploading the rebuilt image: ") + hipGetErrorString(err));
        }
        for (auto &f : fresh) { // every device has the new image: let each finish what it was doing with the old one, then swap
            DeviceScene &d = scene->dev[f.first];
            if (hipSetDevice(f.first) == hipSuccess) { (void) hipDeviceSynchronize(); (void) hipFree(d.image); }
            d.image = f.second;
        }
    }
    out.tuned = 1;
    out.probe_rays = (int32_t) rays.size();
    out.nodes_before = tr.nodesBefore; out.nodes_after = tr.nodesAfter;
    out.box_tests_before = tr.visitsBefore; out.box_tests_after = tr.visitsAfter;
    out.build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return RT_OK;
}

extern "C" {

int rt_scene_tune(rt_scene *scene, const rt_camera *camera, int32_t max_w, int32_t max_h, uint64_t seed, int32_t device, rt_tune_info *info) {
    rt_tune_info out{};
    auto report = [&]() {
        if (!info) return;
        const uint32_t sz = info->struct_size;
        memcpy(info, &out, sz < sizeof(out) ? sz : sizeof(out));
        info->struct_size = sz;
    };
    if (!scene) return fail(RT_ERR_INVALID_ARGUMENT, "scene is NULL");
    if (info && info->struct_size < sizeof(uint32_t)) return fail(RT_ERR_INVALID_ARGUMENT, "rt_tune_info.struct_size is not set");
    int rc = check_geometry(camera, max_w, max_h, 0, 1, 0);
    if (rc != RT_OK) return rc;
    rth::HostScene &h = scene->host;
    out.nodes_before = out.nodes_after = h.off.n_nodes;
    if (h.walkKind == RT_WALK_TREE_REFERENCE || h.nBounded < 3) { report(); return RT_OK; }
    DeviceGuard guard;
    rc = guard.enter(device);
    if (rc != RT_OK) return rc;

    // ---- the probe: 16 rows spread over the frame at a reduced sample count (a few million rays), traced by the TIMED kernel
    // variant, rays logged 1 in 2^k ----
    const int rows = 2 * max_h + 1, cols = 2 * max_w + 1;
    const int nProbe = rows < 16 ? rows : 16, stride = rows / nProbe, first = stride / 2;
    const uint32_t cap = 1u << 15, want = 8192u;
    rt_camera probeCam = *camera;
    {
        const double perSample = (double) nProbe * (double) cols * 3.3; // rays one sample of every probe pixel brings
        int spp = (int) std::ceil(4.0e6 / perSample);
        if (spp < 48) spp = 48; // well past the 2k+1 = 11 samples every pixel gets: the pixels that go on (Scene.fs:185-192) must weigh in the
                                // probe as they do in the frame, or the tree is tuned for the sky's rays
        if (spp < camera->samples_per_pixel) probeCam.samples_per_pixel = spp;
    }
    const double upper = (double) nProbe * (double) cols * (double) probeCam.samples_per_pixel * 4.0; // ~4 rays per sample
    int k = 0;
    while (k < 20 && upper / (double) (1u << k) > (double) (cap / 2u)) ++k;
    struct Bufs {
        unsigned char *all = nullptr;
        ~Bufs() { (void) hipFree(all); }
    } b;
    const size_t accumBytes = ((size_t) nProbe * (size_t) cols * 16u + 255u) & ~(size_t) 255u, rayBytes = (size_t) cap * 48u;
    HIP_TRY(hipMalloc((void **) &b.all, accumBytes + rayBytes + 256u));
    int32_t *pAccum = (int32_t *) b.all;
    double *pRays = (double *) (b.all + accumBytes);
    unsigned int *pCount = (unsigned int *) (b.all + accumBytes + rayBytes);
    unsigned int logged = 0;
    for (int attempt = 0; attempt < 12; ++attempt) {
        HIP_TRY(hipMemsetAsync(pCount, 0, sizeof(unsigned int), nullptr));
        const RayLog log{pRays, pCount, cap, (1u << k) - 1u};
        Pending pd;
        rt_render_options po{};
        po.struct_size = (uint32_t) sizeof(po);
        po.chunk_pixels = 4; // 16 rows are a few thousand pixels for 4096 waves: small units, or most waves get none and a few get the long ones
        po.passes = 1;       // the fused kernel: the only mode whose instantiations carry the ray log (rt_render_kernel.h, Sched<.., LOG>)
        rc = launch_render(scene, &probeCam, max_w, max_h, seed, device, first, stride, nProbe, 0u, pAccum, nullptr, nullptr, &po, true, pd, &log);
        if (rc != RT_OK) return rc;
        rt_stats st;
        rc = collect_stats(pd, &st);
        if (rc != RT_OK) return rc;
        out.probe_ms += st.kernel_ms;
        HIP_TRY(hipMemcpy(&logged, pCount, sizeof(logged), hipMemcpyDeviceToHost));
        if (logged > cap && k < 20) { k += 2; continue; }           // too many: log more thinly
        if (logged < want / 8u && k > 0) { k = k > 3 ? k - 3 : 0; continue; } // too few: log more densely
        break;
    }
    out.probe_rows = nProbe;
    // A log that still overflows holds whichever rays happened to be appended first: a scheduling-dependent subset, from which the
    // tree (and the box-test counters) would differ from run to run and from rank to rank.  Pixels would not, but "the same call
    // yields the same tree" is part of the contract: such a scene is left untuned.
    if (logged > cap || logged == 0) { report(); return RT_OK; }
    // ---- host: sort the log (its order depends on scheduling, its content does not) and rebuild ----
    std::vector<RawRay> raw(logged);
    HIP_TRY(hipMemcpy(raw.data(), pRays, (size_t) logged * sizeof(RawRay), hipMemcpyDeviceToHost));
    std::sort(raw.begin(), raw.end(), [](const RawRay &x, const RawRay &y) { return memcmp(&x, &y, sizeof(RawRay)) < 0; });
    rc = apply_tune(scene, raw, out);
    if (rc == RT_OK) report();
    return rc;
}

int rt_scene_tune_rays(rt_scene *scene, const double *rays, size_t n_rays, rt_tune_info *info) {
    rt_tune_info out{};
    if (!scene) return fail(RT_ERR_INVALID_ARGUMENT, "scene is NULL");
    if (n_rays > 0 && !rays) return fail(RT_ERR_INVALID_ARGUMENT, "rays is NULL");
    if (info && info->struct_size < sizeof(uint32_t)) return fail(RT_ERR_INVALID_ARGUMENT, "rt_tune_info.struct_size is not set");
    out.nodes_before = out.nodes_after = scene->host.off.n_nodes;
    std::vector<RawRay> raw(n_rays);
    if (n_rays) memcpy(raw.data(), rays, n_rays * sizeof(RawRay));
    const int rc = apply_tune(scene, raw, out);
    if (rc == RT_OK && info) {
        const uint32_t sz = info->struct_size;
        memcpy(info, &out, sz < sizeof(out) ? sz : sizeof(out));
        info->struct_size = sz;
    }
    return rc;
}

int rt_render(const rt_scene *scene, const rt_camera *camera, int32_t max_w, int32_t max_h, uint64_t seed, int32_t device,
              int32_t row_first, int32_t row_stride, int32_t n_rows, uint32_t flags, int32_t *accum_host, uint8_t *rgb_host, rt_stats *stats) {
    int rc = check_geometry(camera, max_w, max_h, row_first, row_stride, n_rows);
    if (rc != RT_OK) return rc;
    if (n_rows > 0 && !accum_host) return fail(RT_ERR_INVALID_ARGUMENT, "accum_host is NULL");
    DeviceGuard guard;
    rc = guard.enter(device);
    if (rc != RT_OK) return rc;
    const auto t0 = std::chrono::steady_clock::now();
    const size_t npx = (size_t) n_rows * (size_t) (2 * max_w + 1);
    int32_t *d_accum = nullptr;
    uint8_t *d_rgb = nullptr;
    if (npx > 0) {
        HIP_TRY(hipMalloc((void **) &d_accum, npx * 16u));
        if (rgb_host) {
            hipError_t e = hipMalloc((void **) &d_rgb, npx * 3u);
            if (e != hipSuccess) { (void) hipFree(d_accum); return fail(RT_ERR_HIP, std::string("hipMalloc rgb: ") + hipGetErrorString(e)); }
        }
    }
    rt_stats local;
    rc = rt_render_device(scene, camera, max_w, max_h, seed, device, row_first, row_stride, n_rows, flags, d_accum, d_rgb, nullptr, &local);
    if (rc == RT_OK && npx > 0) {
        hipError_t e = hipMemcpy(accum_host, d_accum, npx * 16u, hipMemcpyDeviceToHost);
        if (e == hipSuccess && rgb_host) e = hipMemcpy(rgb_host, d_rgb, npx * 3u, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(RT_ERR_HIP, std::string("hipMemcpy: ") + hipGetErrorString(e));
    }
    (void) hipFree(d_accum);
    (void) hipFree(d_rgb);
    if (rc == RT_OK && stats) {
        *stats = local;
        stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return rc;
}

} // extern "C"

// ------------------------------------------------------------------------------------------------------------
// rt_render_frame: one process, several devices, one gather
// ------------------------------------------------------------------------------------------------------------
namespace {

// librccl.so is loaded on first use (dlopen), so the library itself depends on libamdhip64 only.
struct RcclApi {
    typedef int (*init_all_t)(void **, int, const int *);
    typedef int (*destroy_t)(void *);
    typedef int (*group_t)(void);
    typedef int (*send_t)(const void *, size_t, int, int, void *, hipStream_t);
    typedef int (*recv_t)(void *, size_t, int, int, void *, hipStream_t);
    typedef const char *(*errstr_t)(int);
    void *lib = nullptr;
    init_all_t CommInitAll = nullptr;
    destroy_t CommDestroy = nullptr;
    group_t GroupStart = nullptr, GroupEnd = nullptr;
    send_t Send = nullptr;
    recv_t Recv = nullptr;
    errstr_t GetErrorString = nullptr;
    std::string why;
    bool ok() const { return lib != nullptr; }
};
static RcclApi load_rccl() {
    RcclApi r;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names) { h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
    if (!h) { r.why = std::string("librccl.so could not be loaded: ") + (dlerror() ? dlerror() : "?"); return r; }
    r.CommInitAll = (RcclApi::init_all_t) dlsym(h, "ncclCommInitAll");
    r.CommDestroy = (RcclApi::destroy_t) dlsym(h, "ncclCommDestroy");
    r.GroupStart = (RcclApi::group_t) dlsym(h, "ncclGroupStart");
    r.GroupEnd = (RcclApi::group_t) dlsym(h, "ncclGroupEnd");
    r.Send = (RcclApi::send_t) dlsym(h, "ncclSend");
    r.Recv = (RcclApi::recv_t) dlsym(h, "ncclRecv");
    r.GetErrorString = (RcclApi::errstr_t) dlsym(h, "ncclGetErrorString");
    if (!r.CommInitAll || !r.CommDestroy || !r.GroupStart || !r.GroupEnd || !r.Send || !r.Recv || !r.GetErrorString) {
        r.why = "librccl.so lacks a needed symbol";
        return r;
    }
    r.lib = h;
    return r;
}
static RcclApi &rccl() { static RcclApi api = load_rccl(); return api; }
#define RT_NCCL_INT32 2 /* ncclInt32 (rccl.h) */

// communicators are expensive to set up (~0.1-1 s): kept per device list for the life of the process
static std::mutex g_comm_mu;
static std::map<std::vector<int>, std::vector<void *>> g_comms;
static int comms_for(const std::vector<int> &devs, std::vector<void *> **out) {
    std::lock_guard<std::mutex> lock(g_comm_mu);
    auto it = g_comms.find(devs);
    if (it == g_comms.end()) {
        std::vector<void *> c(devs.size(), nullptr);
        const int rc = rccl().CommInitAll(c.data(), (int) devs.size(), devs.data());
        if (rc != 0) return fail(RT_ERR_HIP, std::string("ncclCommInitAll: ") + rccl().GetErrorString(rc));
        it = g_comms.emplace(devs, c).first;
    }
    *out = &it->second;
    return RT_OK;
}

struct FrameDevice { // one device's share of a frame; everything is released on every exit path
    int device = -1;
    hipStream_t stream = nullptr;
    int32_t *accum = nullptr; // this device's shard
    int32_t *stage = nullptr; // on devices[0]: where this device's shard arrives
    int32_t n_rows = 0;
    Pending pd;
    ~FrameDevice() {
        if (device < 0) return;
        if (hipSetDevice(device) != hipSuccess) return;
        pd.release();
        if (stream) { (void) hipStreamSynchronize(stream); (void) hipStreamDestroy(stream); }
        (void) hipFree(accum);
    }
};

} // namespace

extern "C" {

int rt_render_frame(const rt_scene *scene, const rt_camera *camera, int32_t max_w, int32_t max_h, uint64_t seed, const int32_t *devices,
                    int32_t n_devices, uint32_t flags, int32_t gather, const rt_render_options *options, int32_t *accum_host, uint8_t *rgb_host,
                    rt_stats *stats) {
    if (!scene) return fail(RT_ERR_INVALID_ARGUMENT, "scene is NULL");
    if (!devices || n_devices < 1 || n_devices > 64) return fail(RT_ERR_INVALID_ARGUMENT, "devices: 1 to 64 device ids");
    if (gather < RT_GATHER_AUTO || gather > RT_GATHER_HOST) return fail(RT_ERR_INVALID_ARGUMENT, "gather must be one of RT_GATHER_*");
    int rc = check_geometry(camera, max_w, max_h, 0, 1, 2 * max_h + 1);
    if (rc != RT_OK) return rc;
    if (!accum_host) return fail(RT_ERR_INVALID_ARGUMENT, "accum_host is NULL");
    const int visible = visible_devices();
    if (visible <= 0) return fail(RT_ERR_NO_DEVICE, "no HIP device visible: the render path has no CPU fallback");
    bool distinct = true;
    for (int i = 0; i < n_devices; ++i) {
        if (devices[i] < 0 || devices[i] >= visible) return fail(RT_ERR_INVALID_ARGUMENT, "device index out of range");
        for (int j = 0; j < i; ++j) distinct = distinct && devices[j] != devices[i];
    }
    if (gather == RT_GATHER_RCCL && !distinct) return fail(RT_ERR_INVALID_ARGUMENT, "RT_GATHER_RCCL needs distinct devices (one communicator rank per GPU)");
    if (gather == RT_GATHER_RCCL && !rccl().ok()) return fail(RT_ERR_UNSUPPORTED, rccl().why);
    const bool viaSelf = gather == RT_GATHER_RCCL; // asked for by name: devices[0]'s own shard goes through ncclSend/ncclRecv as well
    if (gather == RT_GATHER_AUTO) gather = (n_devices > 1 && distinct && rccl().ok()) ? RT_GATHER_RCCL : RT_GATHER_PEER;

    const auto t0 = std::chrono::steady_clock::now();
    const int rows = 2 * max_h + 1, cols = 2 * max_w + 1;
    const size_t rowBytes = (size_t) cols * 16u;
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) { (void) hipGetLastError(); prev = -1; }
    struct Restore { int prev; ~Restore() { if (prev >= 0) (void) hipSetDevice(prev); } } restore{prev};

    std::vector<FrameDevice> fd((size_t) n_devices); // destroyed (streams drained, buffers freed) before `restore`
    struct StageOwner { std::vector<int32_t *> bufs; int device; ~StageOwner() { if (hipSetDevice(device) == hipSuccess) for (auto *b : bufs) (void) hipFree(b); } } stages{{}, devices[0]};

    // ---- every device renders its interleaved rows on a stream of its own; nothing waits here ----
    for (int i = 0; i < n_devices; ++i) {
        FrameDevice &f = fd[(size_t) i];
        HIP_TRY(hipSetDevice(devices[i]));
        f.device = devices[i];
        f.n_rows = (rows - i + n_devices - 1) / n_devices;
        if (f.n_rows < 0) f.n_rows = 0;
        HIP_TRY(hipStreamCreateWithFlags(&f.stream, hipStreamNonBlocking));
        if (f.n_rows > 0) HIP_TRY(hipMalloc((void **) &f.accum, (size_t) f.n_rows * rowBytes));
        rc = launch_render(scene, camera, max_w, max_h, seed, devices[i], i, n_devices, f.n_rows, flags, f.accum, nullptr, f.stream, options,
                           stats != nullptr, f.pd);
        if (rc != RT_OK) return rc;
    }

    // ---- one gather ----
    const size_t dpitch = (size_t) n_devices * rowBytes;
    auto to_host = [&](const int32_t *src, int i, hipStream_t st) -> hipError_t { // de-interleaving strided copy: shard row j -> image row i + j*n
        if (fd[(size_t) i].n_rows == 0) return hipSuccess;
        return hipMemcpy2DAsync((char *) accum_host + (size_t) i * rowBytes, dpitch, src, rowBytes, rowBytes, (size_t) fd[(size_t) i].n_rows, hipMemcpyDeviceToHost, st);
    };
    if (gather == RT_GATHER_HOST) {
        for (int i = 0; i < n_devices; ++i) {
            HIP_TRY(hipSetDevice(devices[i]));
            HIP_TRY(to_host(fd[(size_t) i].accum, i, fd[(size_t) i].stream));
        }
        for (int i = 0; i < n_devices; ++i) { HIP_TRY(hipSetDevice(devices[i])); HIP_TRY(hipStreamSynchronize(fd[(size_t) i].stream)); }
    } else {
        HIP_TRY(hipSetDevice(devices[0]));
        for (int i = 0; i < n_devices; ++i) {
            int32_t *b = nullptr;
            if ((i > 0 || viaSelf) && fd[(size_t) i].n_rows > 0) HIP_TRY(hipMalloc((void **) &b, (size_t) fd[(size_t) i].n_rows * rowBytes));
            stages.bufs.push_back(b);
            fd[(size_t) i].stage = b;
        }
        if (gather == RT_GATHER_PEER) {
            for (int i = 1; i < n_devices; ++i) {
                if (fd[(size_t) i].n_rows == 0) continue;
                HIP_TRY(hipSetDevice(devices[i]));
                HIP_TRY(hipMemcpyPeerAsync(fd[(size_t) i].stage, devices[0], fd[(size_t) i].accum, devices[i], (size_t) fd[(size_t) i].n_rows * rowBytes, fd[(size_t) i].stream));
            }
            for (int i = 1; i < n_devices; ++i) { HIP_TRY(hipSetDevice(devices[i])); HIP_TRY(hipStreamSynchronize(fd[(size_t) i].stream)); }
        } else { // RCCL over xGMI: every sender on its own render stream, every receive on devices[0]'s
            std::vector<int> devs(devices, devices + n_devices);
            std::vector<void *> *comms = nullptr;
            rc = comms_for(devs, &comms);
            if (rc != RT_OK) return rc;
            RcclApi &nc = rccl();
            // nothing returns between GroupStart and GroupEnd: an early return would leave RCCL in group mode for the rest of the
            // process (the communicators are cached); errors are collected and reported after the group has been closed
            int e = nc.GroupStart();
            hipError_t he = hipSuccess;
            for (int i = viaSelf ? 0 : 1; i < n_devices && e == 0 && he == hipSuccess; ++i) {
                const size_t count = (size_t) fd[(size_t) i].n_rows * (size_t) cols * 4u;
                if (count == 0) continue;
                he = hipSetDevice(devices[i]);
                if (he != hipSuccess) break;
                e = nc.Send(fd[(size_t) i].accum, count, RT_NCCL_INT32, 0, (*comms)[(size_t) i], fd[(size_t) i].stream);
                if (e != 0) break;
                he = hipSetDevice(devices[0]);
                if (he != hipSuccess) break;
                e = nc.Recv(fd[(size_t) i].stage, count, RT_NCCL_INT32, i, (*comms)[0], fd[0].stream);
            }
            const int e2 = nc.GroupEnd();
            if (e == 0) e = e2;
            if (he != hipSuccess) return fail(RT_ERR_HIP, std::string("RCCL gather: hipSetDevice: ") + hipGetErrorString(he));
            if (e != 0) return fail(RT_ERR_HIP, std::string("RCCL gather: ") + nc.GetErrorString(e));
            for (int i = 1; i < n_devices; ++i) { HIP_TRY(hipSetDevice(devices[i])); HIP_TRY(hipStreamSynchronize(fd[(size_t) i].stream)); }
        }
        HIP_TRY(hipSetDevice(devices[0]));
        for (int i = 0; i < n_devices; ++i) HIP_TRY(to_host(fd[(size_t) i].stage ? fd[(size_t) i].stage : fd[(size_t) i].accum, i, fd[0].stream));
        HIP_TRY(hipStreamSynchronize(fd[0].stream));
    }

    if (rgb_host) { // PixelStats.mean (Pixel.fs:103-108): integer division of the sums by Count
        const size_t npx = (size_t) rows * (size_t) cols;
        for (size_t i = 0; i < npx; ++i) {
            const int32_t *a = accum_host + i * 4;
            rgb_host[i * 3 + 0] = (uint8_t) (a[1] / a[0]);
            rgb_host[i * 3 + 1] = (uint8_t) (a[2] / a[0]);
            rgb_host[i * 3 + 2] = (uint8_t) (a[3] / a[0]);
        }
    }
    if (stats) {
        for (int i = 0; i < n_devices; ++i) {
            HIP_TRY(hipSetDevice(devices[i]));
            rc = collect_stats(fd[(size_t) i].pd, &stats[i]);
            if (rc != RT_OK) return rc;
        }
        const double total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        for (int i = 0; i < n_devices; ++i) stats[i].total_ms = total;
    }
    return RT_OK;
}

// ---- output side ------------------------------------------------------------------------------------------
uint8_t rt_gamma_correct(uint8_t b) { return rth::gamma_correct(b); }

int64_t rt_format_ppm(const uint8_t *rgb, int32_t rows, int32_t cols, int32_t gamma_correct, char *out, size_t out_capacity) {
    if (!rgb || rows <= 0 || cols <= 0) { fail(RT_ERR_INVALID_ARGUMENT, "bad image"); return -RT_ERR_INVALID_ARGUMENT; }
    std::string s = rth::format_ppm(rgb, rows, cols, gamma_correct != 0);
    if (out && out_capacity > s.size()) { memcpy(out, s.data(), s.size()); out[s.size()] = 0; }
    return (int64_t) s.size();
}

int64_t rt_format_pixel_map(const uint8_t *rgb, int32_t rows, int32_t cols, uint8_t *out, size_t out_capacity) {
    if (!rgb || rows <= 0 || cols <= 0) { fail(RT_ERR_INVALID_ARGUMENT, "bad image"); return -RT_ERR_INVALID_ARGUMENT; }
    std::string s = rth::format_pixel_map(rgb, rows, cols);
    if (out && out_capacity >= s.size()) memcpy(out, s.data(), s.size());
    return (int64_t) s.size();
}

int64_t rt_parse_pixel_map(const uint8_t *data, size_t n, int32_t rows, int32_t cols, uint8_t *rgb_out, uint8_t *present_out) {
    if ((!data && n) || rows <= 0 || cols <= 0 || !rgb_out) { fail(RT_ERR_INVALID_ARGUMENT, "bad argument"); return -RT_ERR_INVALID_ARGUMENT; }
    if (present_out) memset(present_out, 0, (size_t) rows * (size_t) cols);
    const int64_t c = rth::parse_pixel_map(data, n, rows, cols, rgb_out, present_out);
    if (c < 0) { fail(RT_ERR_INVALID_ARGUMENT, "pixel map names a pixel outside the image"); return -RT_ERR_INVALID_ARGUMENT; }
    return c;
}

int rt_write_ppm(const char *path, const uint8_t *rgb, int32_t rows, int32_t cols, int32_t gamma_correct) {
    if (!path || !rgb || rows <= 0 || cols <= 0) return fail(RT_ERR_INVALID_ARGUMENT, "bad image or path");
    std::string s = rth::format_ppm(rgb, rows, cols, gamma_correct != 0);
    FILE *f = fopen(path, "wb");
    if (!f) return fail(RT_ERR_IO, std::string("cannot open ") + path);
    const size_t w = fwrite(s.data(), 1, s.size(), f);
    const int c = fclose(f);
    if (w != s.size() || c != 0) return fail(RT_ERR_IO, std::string("short write to ") + path);
    return RT_OK;
}

} // extern "C"

// ============================================================================================================
// Device unit hooks: tiny kernels that call the SAME inlined device functions the render kernel uses.
// ============================================================================================================
namespace {

template <typename T> struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    ~DevBuf() { if (p) (void) hipFree(p); }
    hipError_t alloc(size_t count) { n = count; return count ? hipMalloc((void **) &p, count * sizeof(T)) : hipSuccess; }
    hipError_t up(const T *h) { return n ? hipMemcpy(p, h, n * sizeof(T), hipMemcpyHostToDevice) : hipSuccess; }
    hipError_t down(T *h) { return n ? hipMemcpy(h, p, n * sizeof(T), hipMemcpyDeviceToHost) : hipSuccess; }
};

__global__ void k_float_producer(Rng r, int n, double *out) {
    if (threadIdx.x == 0 && blockIdx.x == 0)
        for (int i = 0; i < n; ++i) out[i] = rng_get(r);
}
__global__ void k_stream_state(uint64_t seedKey, int n, const uint64_t *pixel, const uint32_t *sample, uint32_t *out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Rng r = stream_for(pixel_key(seedKey, pixel[i]), sample[i]);
    out[i * 4] = r.x; out[i * 4 + 1] = r.y; out[i * 4 + 2] = r.z; out[i * 4 + 3] = r.w;
}
__global__ void k_bbox_hits(int n, const double *rays, const double *boxes, int32_t *hit) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double *r = rays + i * 6, *b = boxes + i * 6;
    d2 bx, by, bz;
    bx.x = b[0]; bx.y = b[3]; by.x = b[1]; by.y = b[4]; bz.x = b[2]; bz.y = b[5];
    hit[i] = bbox_hits(1.0 / r[3], 1.0 / r[4], 1.0 / r[5], mk(r[0], r[1], r[2]), bx, by, bz) ? 1 : 0;
}
__global__ void k_sphere_isect(int n, const double *rays, const double *sph, double *t) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double *r = rays + i * 6, *s = sph + i * 4;
    t[i] = sphere_first_intersection(mk(r[0], r[1], r[2]), mk(r[3], r[4], r[5]), mk(s[0], s[1], s[2]), s[3] * s[3]);
}
__global__ void k_plane_isect(int n, const double *rays, const double *pl, double *t) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double *r = rays + i * 6, *p = pl + i * 6;
    t[i] = plane_intersection(mk(r[0], r[1], r[2]), mk(r[3], r[4], r[5]), mk(p[0], p[1], p[2]), mk(p[3], p[4], p[5]));
}
RTD_INLINE uint32_t ld_rgb(const uint8_t *p) { return (uint32_t) p[0] | ((uint32_t) p[1] << 8) | ((uint32_t) p[2] << 16); }
RTD_INLINE void st_rgb(uint8_t *p, uint32_t c) { p[0] = (uint8_t) c; p[1] = (uint8_t) (c >> 8); p[2] = (uint8_t) (c >> 16); }
__global__ void k_pixel_combine(int n, const uint8_t *a, const uint8_t *b, uint8_t *out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    st_rgb(out + i * 3, pix_combine(ld_rgb(a + i * 3), ld_rgb(b + i * 3)));
}
__global__ void k_pixel_darken(int n, const uint8_t *p, const double *albedo, uint8_t *out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    st_rgb(out + i * 3, pix_darken(albedo[i], ld_rgb(p + i * 3)));
}
__global__ void k_arith(int op, int n, const double *a, const double *b, double *out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double r;
    switch (op) {
    case 0: r = 1.0 / a[i]; break;
    case 1: r = sqrt(a[i]); break;
    case 2: r = rint(a[i]); break;
    case 3: r = a[i] / b[i]; break;
    case 4: r = pow5(a[i]); break;
    case 5: r = sqrt_above_tol(a[i]); break;
    case 6: r = inv_sqrt_above_tol(a[i]); break;
    case 7: r = rtt::cr_acos(a[i]); break;
    case 8: r = rtt::cr_sin(a[i]); break;
    default: r = rtt::cr_atan2(a[i], b[i]); break;
    }
    out[i] = r;
}

struct HookScene { RenderParams p; };
static SceneView<false> host_view_params(const DeviceScene *ds, const rth::HostScene &h, RenderParams &p) {
    p = RenderParams{};
    p.off = h.off;
    p.scene_image = ds->image;
    p.tex = ds->tex;
    p.texels = ds->texels;
    return SceneView<false>{};
}

__global__ void k_reflection(const RenderParams p, int n, const int32_t *obj, const double *ray_in, const uint8_t *col_in,
                             const double *strike, uint32_t *rng, int32_t *absorbed, uint8_t *col_out, double *ray_out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const SceneView<false> sc = make_view<false>(p, nullptr);
    const double *r = ray_in + i * 6, *s = strike + i * 3;
    V3 o = mk(r[0], r[1], r[2]), d = mk(r[3], r[4], r[5]);
    uint32_t c = ld_rgb(col_in + i * 3);
    Rng g; g.x = rng[i * 4]; g.y = rng[i * 4 + 1]; g.z = rng[i * 4 + 2]; g.w = rng[i * 4 + 3];
    bool ab = reflection<false>(sc, obj[i], mk(s[0], s[1], s[2]), o, d, c, g);
    absorbed[i] = ab ? 1 : 0;
    st_rgb(col_out + i * 3, c);
    double *ro = ray_out + i * 6;
    ro[0] = o.x; ro[1] = o.y; ro[2] = o.z; ro[3] = d.x; ro[4] = d.y; ro[5] = d.z;
    rng[i * 4] = g.x; rng[i * 4 + 1] = g.y; rng[i * 4 + 2] = g.z; rng[i * 4 + 3] = g.w;
}
__global__ void k_hit_object(const RenderParams p, int n, const double *rays, int32_t *hit, double *strike, uint32_t *counters) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const SceneView<false> sc = make_view<false>(p, nullptr);
    const double *r = rays + i * 6;
    V3 o = mk(r[0], r[1], r[2]), d = mk(r[3], r[4], r[5]);
    Counters cnt; cnt.rays = cnt.aabb = cnt.prim = cnt.refl = 0;
    double t;
    int obj = hit_object<false, true>(sc, o, d, t, cnt);
    hit[i] = obj;
    V3 sp = walk(o, d, t);
    if (strike) {
        const double nanv = __builtin_nan("");
        strike[i * 3] = obj < 0 ? nanv : sp.x; strike[i * 3 + 1] = obj < 0 ? nanv : sp.y; strike[i * 3 + 2] = obj < 0 ? nanv : sp.z;
    }
    if (counters) { counters[i * 2] = cnt.aabb; counters[i * 2 + 1] = cnt.prim; }
}
// Scene.hitObject as the render kernel's timed variant runs it: scene staged into LDS, the hand-written single-precision filter
// loop (node_loop_lds32), the leaf pass with the exact box and sphere tests between its runs, the unbounded objects last.  One ray
// per lane, 1024 rays per workgroup.
__global__ void __launch_bounds__(1024) k_hit_object_lds(const RenderParams p, int n, const double *rays, int32_t *hit, double *strike) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    (void) stage_scene<1024, true>(p, smem);
    const SceneView<true> sc = make_view<true, true>(p, smem);
    const int i = blockIdx.x * 1024 + threadIdx.x;
    const bool valid = i < n;
    const double *r = rays + (size_t) (valid ? i : 0) * 6;
    const V3 o = mk(r[0], r[1], r[2]), d = mk(r[3], r[4], r[5]);
    Walk w;
    walk_begin(w, sc.first);
    if (!valid) w.off = sc.end;
    const WalkCtx32 f = walk_ctx32(o, d, p.off.bmax);
    double bestF = __builtin_inf();
    // the claim that lets the leaf pass skip the exact box test needs unit directions: this hook's rays are arbitrary, so it holds
    // a lane to the claim only when its direction is one (|d|^2 within 1e-15 of 1, as Ray.make' leaves it)
    const double n2 = dot(d, d);
    const bool implied = p.off.box_implied != 0 && n2 >= 1.0 - 1e-15 && n2 <= 1.0 + 1e-15;
    uint32_t pend = 0u;
    for (;;) {
#ifdef RTD_STAGE_CLOCKS
        unsigned nTrips = 0u, nLanes = 0u;
        w.off = node_loop_lds32(w.off, pend, sc.end, 0, f, nTrips, nLanes);
#else
        w.off = node_loop_lds32(w.off, pend, sc.end, 0, f); // until no lane of the wave can step: walks exhausted or queues full
#endif
        if (__builtin_amdgcn_ballot_w64(pend != 0u) == 0ull) break;
        if (pend != 0u) leaf_test_object_exact<true>(sc, o, d, bestF, w, pend_pop(pend), implied);
    }
    Counters cnt; cnt.rays = cnt.aabb = cnt.prim = cnt.refl = 0;
    unbounded_tests<true, false>(sc, o, d, w, cnt);
    if (!valid) return;
    hit[i] = w.best;
    const V3 sp = walk(o, d, w.bestLen);
    const double nanv = __builtin_nan("");
    strike[i * 3] = w.best < 0 ? nanv : sp.x; strike[i * 3 + 1] = w.best < 0 ? nanv : sp.y; strike[i * 3 + 2] = w.best < 0 ? nanv : sp.z;
}
// pixel_candidates (rt_device.h) for n pixels of a camera: the Leaves a pixel's camera rays can reach, as the render kernel computes
// them once per pixel (scene staged into LDS exactly as the timed kernel stages it).  out[i*2], out[i*2+1] = the two queue words.
__global__ void __launch_bounds__(1024) k_pixel_candidates(const RenderParams p, const CameraParams cam, int n, const int32_t *rowcol, uint32_t *out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    (void) stage_scene<1024, true>(p, smem);
    const SceneView<true> sc = make_view<true, true>(p, smem);
    const int i = blockIdx.x * 1024 + threadIdx.x;
    if (i >= n) return;
    uint32_t second = 0u;
    const uint32_t first = pixel_candidates<true, true>(sc, cam, rowcol[i * 2], rowcol[i * 2 + 1], second);
    out[i * 2] = first;
    out[i * 2 + 1] = second;
}
// The single-precision filter of the timed node loop next to the exact test, box by box: out[i] = exact | filter << 1.
// Boxes arrive as (min, max) doubles; the host rounds them outward exactly as the scene image does (rth::f32_down / f32_up).
__global__ void k_bbox_filter(int n, const double *rays, const double *boxes, const float *boxes32, float bmax, int32_t *out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double *r = rays + i * 6, *b = boxes + i * 6;
    const float *f = boxes32 + i * 6;
    d2 bx, by, bz;
    bx.x = b[0]; bx.y = b[3]; by.x = b[1]; by.y = b[4]; bz.x = b[2]; bz.y = b[5];
    const V3 o = mk(r[0], r[1], r[2]), d = mk(r[3], r[4], r[5]);
    const bool exact = bbox_hits(1.0 / d.x, 1.0 / d.y, 1.0 / d.z, o, bx, by, bz);
    const WalkCtx32 c = walk_ctx32(o, d, bmax);
    // the loop's own instruction forms (v_max3 / v_min3 / v_max / v_cmp_nlt): NaN handling is theirs
    const float nx = c.nX ? f[3] : f[0], fx = c.nX ? f[0] : f[3], ny = c.nY ? f[4] : f[1], fy = c.nY ? f[1] : f[4], nz = c.nZ ? f[5] : f[2], fz = c.nZ ? f[2] : f[5];
    float t0 = nx, t1 = fx, t2 = ny, t3 = fy, t4 = nz, t5 = fz;
    unsigned long long m;
    asm volatile("v_fma_f32 %0, %0, %7, %10\n\tv_fma_f32 %1, %1, %7, %13\n\t"
                 "v_fma_f32 %2, %2, %8, %11\n\tv_fma_f32 %3, %3, %8, %14\n\t"
                 "v_fma_f32 %4, %4, %9, %12\n\tv_fma_f32 %5, %5, %9, %15\n\t"
                 "v_max3_f32 %0, %0, %2, %4\n\tv_min3_f32 %1, %1, %3, %5\n\tv_max_f32 %0, 0, %0\n\tv_cmp_nlt_f32_e64 %6, %1, %0"
                 : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3), "+v"(t4), "+v"(t5), "=s"(m)
                 : "v"(c.ix), "v"(c.iy), "v"(c.iz), "v"(c.cnx), "v"(c.cny), "v"(c.cnz), "v"(c.cfx), "v"(c.cfy), "v"(c.cfz));
    const bool filter = (m >> (threadIdx.x & 63)) & 1ull;
    out[i] = (exact ? 1 : 0) | (filter ? 2 : 0) | (bbox_filter(c, f[0], f[3], f[1], f[4], f[2], f[5]) ? 4 : 0);
}
__global__ void k_trace_ray(const RenderParams p, int depth, int n, const double *rays, uint32_t *rng, uint8_t *col_out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const SceneView<false> sc = make_view<false>(p, nullptr);
    const double *r = rays + i * 6;
    Rng g; g.x = rng[i * 4]; g.y = rng[i * 4 + 1]; g.z = rng[i * 4 + 2]; g.w = rng[i * 4 + 3];
    Counters cnt; cnt.rays = cnt.aabb = cnt.prim = cnt.refl = 0;
    uint32_t c = trace_ray<false, false>(sc, depth, mk(r[0], r[1], r[2]), mk(r[3], r[4], r[5]), g, cnt);
    st_rgb(col_out + i * 3, c);
    rng[i * 4] = g.x; rng[i * 4 + 1] = g.y; rng[i * 4 + 2] = g.z; rng[i * 4 + 3] = g.w;
}
__global__ void k_texture(const RenderParams p, int tex, int n, const double *pts, double *uv, uint8_t *col) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double u[2] = {__builtin_nan(""), __builtin_nan("")};
    uint32_t c = texture_colour_at(p.tex, p.texels, tex, mk(pts[i * 3], pts[i * 3 + 1], pts[i * 3 + 2]), u);
    if (uv) { uv[i * 2] = u[0]; uv[i * 2 + 1] = u[1]; }
    st_rgb(col + i * 3, c);
}

static inline unsigned blocks_for(int n) { return (unsigned) ((n + 255) / 256); }

static int hook_scene_params(DeviceGuard &guard, int device, const rt_scene *scene, RenderParams &p) {
    if (!scene) return fail(RT_ERR_INVALID_ARGUMENT, "scene is NULL");
    int rc = guard.enter(device);
    if (rc != RT_OK) return rc;
    DeviceScene *ds = nullptr;
    rc = device_scene(const_cast<rt_scene *>(scene), device, &ds);
    if (rc != RT_OK) return rc;
    (void) host_view_params(ds, scene->host, p);
    return RT_OK;
}

} // namespace

extern "C" {

int rt_dev_float_producer(int32_t device, const uint32_t state[4], int32_t n, double *out) {
    if (!state || !out || n < 0) return fail(RT_ERR_INVALID_ARGUMENT, "bad argument");
    DeviceGuard guard;
    int rc = guard.enter(device);
    if (rc != RT_OK) return rc;
    DevBuf<double> d;
    HIP_TRY(d.alloc((size_t) n));
    Rng r; r.x = state[0]; r.y = state[1]; r.z = state[2]; r.w = state[3];
    if (n) hipLaunchKernelGGL(k_float_producer, dim3(1), dim3(64), 0, 0, r, n, d.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(d.down(out));
    return RT_OK;
}

int rt_dev_stream_state(int32_t device, uint64_t seed, int32_t n, const uint64_t *pixel, const uint32_t *sample, uint32_t *state_out) {
    if (!pixel || !sample || !state_out || n < 0) return fail(RT_ERR_INVALID_ARGUMENT, "bad argument");
    DeviceGuard guard;
    int rc = guard.enter(device);
    if (rc != RT_OK) return rc;
    DevBuf<uint64_t> dp; DevBuf<uint32_t> dsm, dout;
    HIP_TRY(dp.alloc((size_t) n)); HIP_TRY(dsm.alloc((size_t) n)); HIP_TRY(dout.alloc((size_t) n * 4));
    HIP_TRY(dp.up(pixel)); HIP_TRY(dsm.up(sample));
    if (n) hipLaunchKernelGGL(k_stream_state, dim3(blocks_for(n)), dim3(256), 0, 0, mix64(seed + 0x9E3779B97F4A7C15ull), n, dp.p, dsm.p, dout.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(dout.down(state_out));
    return RT_OK;
}

int rt_dev_bbox_hits(int32_t device, int32_t n, const double *rays, const double *boxes, int32_t *hit_out) {
    if (!rays || !boxes || !hit_out || n < 0) return fail(RT_ERR_INVALID_ARGUMENT, "bad argument");
    DeviceGuard guard;
    int rc = guard.enter(device);
    if (rc != RT_OK) return rc;
    DevBuf<double> dr, db; DevBuf<int32_t> dh;
    HIP_TRY(dr.alloc((size_t) n * 6)); HIP_TRY(db.alloc((size_t) n * 6)); HIP_TRY(dh.alloc((size_t) n));
    HIP_TRY(dr.up(rays)); HIP_TRY(db.up(boxes));
    if (n) hipLaunchKernelGGL(k_bbox_hits, dim3(blocks_for(n)), dim3(256), 0, 0, n, dr.p, db.p, dh.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(dh.down(hit_out));
    return RT_OK;
}

int rt_dev_bbox_filter(int32_t device, int32_t n, const double *rays, const double *boxes, double bmax, int32_t *out) {
    if (!rays || !boxes || !out || n < 0) return fail(RT_ERR_INVALID_ARGUMENT, "bad argument");
    DeviceGuard guard;
    int rc = guard.enter(device);
    if (rc != RT_OK) return rc;
    std::vector<float> b32((size_t) n * 6);
    float bm = 1e-30f; // as encode_image: >= every |coordinate| of the batch's rounded boxes, unless the caller names a larger scale
    for (int i = 0; i < n; ++i)
        for (int a = 0; a < 3; ++a) {
            const float lo = rth::f32_down(boxes[(size_t) i * 6 + a]), hi = rth::f32_up(boxes[(size_t) i * 6 + 3 + a]);
            b32[(size_t) i * 6 + a] = lo; b32[(size_t) i * 6 + 3 + a] = hi;
            if (std::fabs(lo) > bm) bm = std::fabs(lo);
            if (std::fabs(hi) > bm) bm = std::fabs(hi);
        }
    if (bmax > (double) bm) bm = rth::f32_up(bmax);
    DevBuf<double> dr, db; DevBuf<float> df; DevBuf<int32_t> dh;
    HIP_TRY(dr.alloc((size_t) n * 6)); HIP_TRY(db.alloc((size_t) n * 6)); HIP_TRY(df.alloc((size_t) n * 6)); HIP_TRY(dh.alloc((size_t) n));
    HIP_TRY(dr.up(rays)); HIP_TRY(db.up(boxes)); HIP_TRY(df.up(b32.data()));
    if (n) hipLaunchKernelGGL(k_bbox_filter, dim3(blocks_for(n)), dim3(256), 0, 0, n, dr.p, db.p, df.p, bm, dh.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(dh.down(out));
    return RT_OK;
}

int rt_dev_sphere_first_intersection(int32_t device, int32_t n, const double *rays, const double *spheres, double *t_out) {
    if (!rays || !spheres || !t_out || n < 0) return fail(RT_ERR_INVALID_ARGUMENT, "bad argument");
    DeviceGuard guard;
    int rc = guard.enter(device);
    if (rc != RT_OK) return rc;
    DevBuf<double> dr, dsph, dt;
    HIP_TRY(dr.alloc((size_t) n * 6)); HIP_TRY(dsph.alloc((size_t) n * 4)); HIP_TRY(dt.alloc((size_t) n));
    HIP_TRY(dr.up(rays)); HIP_TRY(dsph.up(spheres));
    if (n) hipLaunchKernelGGL(k_sphere_isect, dim3(blocks_for(n)), dim3(256), 0, 0, n, dr.p, dsph.p, dt.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(dt.down(t_out));
    return RT_OK;
}

int rt_dev_plane_intersection(int32_t device, int32_t n, const double *rays, const double *planes, double *t_out) {
    if (!rays || !planes || !t_out || n < 0) return fail(RT_ERR_INVALID_ARGUMENT, "bad argument");
    DeviceGuard guard;
    int rc = guard.enter(device);
    if (rc != RT_OK) return rc;
    DevBuf<double> dr, dpl, dt;
    HIP_TRY(dr.alloc((size_t) n * 6)); HIP_TRY(dpl.alloc((size_t) n * 6)); HIP_TRY(dt.alloc((size_t) n));
    HIP_TRY(dr.up(rays)); HIP_TRY(dpl.up(planes));
    if (n) hipLaunchKernelGGL(k_plane_isect, dim3(blocks_for(n)), dim3(256), 0, 0, n, dr.p, dpl.p, dt.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(dt.down(t_out));
    return RT_OK;
}

int rt_dev_pixel_combine(int32_t device, int32_t n, const uint8_t *a, const uint8_t *b, uint8_t *out) {
    if (!a || !b || !out || n < 0) return fail(RT_ERR_INVALID_ARGUMENT, "bad argument");
    DeviceGuard guard;
    int rc = guard.enter(device);
    if (rc != RT_OK) return rc;
    DevBuf<uint8_t> da, db, dout;
    HIP_TRY(da.alloc((size_t) n * 3)); HIP_TRY(db.alloc((size_t) n * 3)); HIP_TRY(dout.alloc((size_t) n * 3));
    HIP_TRY(da.up(a)); HIP_TRY(db.up(b));
    if (n) hipLaunchKernelGGL(k_pixel_combine, dim3(blocks_for(n)), dim3(256), 0, 0, n, da.p, db.p, dout.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(dout.down(out));
    return RT_OK;
}

int rt_dev_pixel_darken(int32_t device, int32_t n, const uint8_t *p, const double *albedo, uint8_t *out) {
    if (!p || !albedo || !out || n < 0) return fail(RT_ERR_INVALID_ARGUMENT, "bad argument");
    DeviceGuard guard;
    int rc = guard.enter(device);
    if (rc != RT_OK) return rc;
    DevBuf<uint8_t> dp, dout; DevBuf<double> da;
    HIP_TRY(dp.alloc((size_t) n * 3)); HIP_TRY(da.alloc((size_t) n)); HIP_TRY(dout.alloc((size_t) n * 3));
    HIP_TRY(dp.up(p)); HIP_TRY(da.up(albedo));
    if (n) hipLaunchKernelGGL(k_pixel_darken, dim3(blocks_for(n)), dim3(256), 0, 0, n, dp.p, da.p, dout.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(dout.down(out));
    return RT_OK;
}

int rt_dev_arith(int32_t device, int32_t op, int32_t n, const double *a, const double *b, double *out) {
    if (!a || !out || n < 0 || op < 0 || op > 9 || ((op == 3 || op == 9) && !b)) return fail(RT_ERR_INVALID_ARGUMENT, "bad argument");
    DeviceGuard guard;
    int rc = guard.enter(device);
    if (rc != RT_OK) return rc;
    DevBuf<double> da, db, dout;
    HIP_TRY(da.alloc((size_t) n)); HIP_TRY(db.alloc(b ? (size_t) n : 0)); HIP_TRY(dout.alloc((size_t) n));
    HIP_TRY(da.up(a));
    if (b) HIP_TRY(db.up(b));
    if (n) hipLaunchKernelGGL(k_arith, dim3(blocks_for(n)), dim3(256), 0, 0, op, n, da.p, db.p, dout.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(dout.down(out));
    return RT_OK;
}

int rt_dev_reflection(int32_t device, const rt_scene *scene, int32_t n, const int32_t *index, const double *ray_in, const uint8_t *colour_in,
                      const double *strike, uint32_t *rng_state, int32_t *absorbed, uint8_t *colour_out, double *ray_out) {
    if (!index || !ray_in || !colour_in || !strike || !rng_state || !absorbed || !colour_out || !ray_out || n < 0)
        return fail(RT_ERR_INVALID_ARGUMENT, "bad argument");
    RenderParams p;
    DeviceGuard guard;
    int rc = hook_scene_params(guard, device, scene, p);
    if (rc != RT_OK) return rc;
    std::vector<int32_t> obj((size_t) n);
    for (int i = 0; i < n; ++i) {
        if (index[i] < 0 || (size_t) index[i] >= scene->host.origToObj.size()) return fail(RT_ERR_INVALID_ARGUMENT, "hittable index out of range");
        obj[(size_t) i] = scene->host.origToObj[(size_t) index[i]];
    }
    DevBuf<int32_t> dobj, dab; DevBuf<double> dri, dst, dro; DevBuf<uint8_t> dci, dco; DevBuf<uint32_t> drng;
    HIP_TRY(dobj.alloc((size_t) n)); HIP_TRY(dab.alloc((size_t) n)); HIP_TRY(dri.alloc((size_t) n * 6)); HIP_TRY(dst.alloc((size_t) n * 3));
    HIP_TRY(dro.alloc((size_t) n * 6)); HIP_TRY(dci.alloc((size_t) n * 3)); HIP_TRY(dco.alloc((size_t) n * 3)); HIP_TRY(drng.alloc((size_t) n * 4));
    HIP_TRY(dobj.up(obj.data())); HIP_TRY(dri.up(ray_in)); HIP_TRY(dst.up(strike)); HIP_TRY(dci.up(colour_in)); HIP_TRY(drng.up(rng_state));
    if (n) hipLaunchKernelGGL(k_reflection, dim3(blocks_for(n)), dim3(256), 0, 0, p, n, dobj.p, dri.p, dci.p, dst.p, drng.p, dab.p, dco.p, dro.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(dab.down(absorbed)); HIP_TRY(dco.down(colour_out)); HIP_TRY(dro.down(ray_out)); HIP_TRY(drng.down(rng_state));
    return RT_OK;
}

int rt_dev_hit_object(int32_t device, const rt_scene *scene, int32_t n, const double *rays, int32_t *hit_index, double *strike, uint32_t *counters) {
    if (!rays || !hit_index || n < 0) return fail(RT_ERR_INVALID_ARGUMENT, "bad argument");
    RenderParams p;
    DeviceGuard guard;
    int rc = hook_scene_params(guard, device, scene, p);
    if (rc != RT_OK) return rc;
    DevBuf<double> dr, dsk; DevBuf<int32_t> dh; DevBuf<uint32_t> dc;
    HIP_TRY(dr.alloc((size_t) n * 6)); HIP_TRY(dh.alloc((size_t) n)); HIP_TRY(dsk.alloc(strike ? (size_t) n * 3 : 0)); HIP_TRY(dc.alloc(counters ? (size_t) n * 2 : 0));
    HIP_TRY(dr.up(rays));
    if (n) hipLaunchKernelGGL(k_hit_object, dim3(blocks_for(n)), dim3(256), 0, 0, p, n, dr.p, dh.p, dsk.p, dc.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(dh.down(hit_index));
    for (int i = 0; i < n; ++i) if (hit_index[i] >= 0) hit_index[i] = scene->host.objToOrig[(size_t) hit_index[i]];
    if (strike) HIP_TRY(dsk.down(strike));
    if (counters) HIP_TRY(dc.down(counters));
    return RT_OK;
}

int rt_dev_hit_object_lds(int32_t device, const rt_scene *scene, int32_t n, const double *rays, int32_t *hit_index, double *strike) {
    if (!rays || !hit_index || !strike || n < 0) return fail(RT_ERR_INVALID_ARGUMENT, "bad argument");
    RenderParams p;
    DeviceGuard guard;
    int rc = hook_scene_params(guard, device, scene, p);
    if (rc != RT_OK) return rc;
    const size_t ldsBytes = scene->host.off.lds32_total;
    if (ldsBytes > RT_LDS_BYTES || scene->host.nBounded + scene->host.nUnbounded >= 16384u) return fail(RT_ERR_UNSUPPORTED, "the scene does not fit the LDS");
    DevBuf<double> dr, dsk; DevBuf<int32_t> dh;
    HIP_TRY(dr.alloc((size_t) n * 6)); HIP_TRY(dh.alloc((size_t) n)); HIP_TRY(dsk.alloc((size_t) n * 3));
    HIP_TRY(dr.up(rays));
    HIP_TRY(hipFuncSetAttribute((const void *) k_hit_object_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int) ldsBytes));
    if (n) hipLaunchKernelGGL(k_hit_object_lds, dim3((unsigned) ((n + 1023) / 1024)), dim3(1024), ldsBytes, 0, p, n, dr.p, dh.p, dsk.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(dh.down(hit_index));
    for (int i = 0; i < n; ++i) if (hit_index[i] >= 0) hit_index[i] = scene->host.objToOrig[(size_t) hit_index[i]];
    HIP_TRY(dsk.down(strike));
    return RT_OK;
}

int rt_dev_pixel_candidates(int32_t device, const rt_scene *scene, const rt_camera *camera, int32_t max_w, int32_t max_h, int32_t n,
                            const int32_t *row_col, int32_t *leaves_out) {
    if (!camera || !row_col || !leaves_out || n < 0 || max_w <= 0 || max_h <= 0) return fail(RT_ERR_INVALID_ARGUMENT, "bad argument");
    RenderParams p;
    DeviceGuard guard;
    int rc = hook_scene_params(guard, device, scene, p);
    if (rc != RT_OK) return rc;
    const size_t ldsBytes = scene->host.off.lds32_total;
    if (ldsBytes > RT_LDS_BYTES || scene->host.nBounded + scene->host.nUnbounded >= 16384u) return fail(RT_ERR_UNSUPPORTED, "the scene does not fit the LDS");
    CameraParams cam{};
    for (int a = 0; a < 3; ++a) { cam.eye[a] = camera->view_origin[a]; cam.xo[a] = camera->xaxis_origin[a]; cam.xd[a] = camera->xaxis_dir[a]; cam.yd[a] = camera->yaxis_dir[a]; }
    cam.vw = camera->viewport_width; cam.vh = camera->viewport_height; cam.max_w = max_w; cam.max_h = max_h;
    DevBuf<int32_t> drc; DevBuf<uint32_t> dout;
    HIP_TRY(drc.alloc((size_t) n * 2)); HIP_TRY(dout.alloc((size_t) n * 2));
    HIP_TRY(drc.up(row_col));
    HIP_TRY(hipFuncSetAttribute((const void *) k_pixel_candidates, hipFuncAttributeMaxDynamicSharedMemorySize, (int) ldsBytes));
    if (n) hipLaunchKernelGGL(k_pixel_candidates, dim3((unsigned) ((n + 1023) / 1024)), dim3(1024), ldsBytes, 0, p, cam, n, drc.p, dout.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    std::vector<uint32_t> w((size_t) n * 2);
    HIP_TRY(dout.down(w.data()));
    // decode: up to four hittable indices per pixel (-1 = none); leaves_out[i*4] = -2 when the pixel's camera rays walk the tree
    for (int i = 0; i < n; ++i) {
        int32_t *o = leaves_out + (size_t) i * 4;
        o[0] = o[1] = o[2] = o[3] = -1;
        if (w[(size_t) i * 2] == RTD_CAND_WALK) { o[0] = -2; continue; }
        int k = 0;
        for (int h = 0; h < 2; ++h) {
            const uint32_t word = w[(size_t) i * 2 + (size_t) h];
            for (int half = 0; half < 2; ++half) {
                const uint32_t e = half == 0 ? (word & 0xFFFFu) : (word >> 16);
                if (e != 0u && k < 4) o[k++] = scene->host.objToOrig[(size_t) (e & (RTD_PEND_MARK - 1u))];
            }
        }
    }
    return RT_OK;
}

int rt_dev_trace_ray(int32_t device, const rt_scene *scene, int32_t bounce_depth, int32_t n, const double *rays, uint32_t *rng_state, uint8_t *colour_out) {
    if (!rays || !rng_state || !colour_out || n < 0 || bounce_depth < 0) return fail(RT_ERR_INVALID_ARGUMENT, "bad argument");
    RenderParams p;
    DeviceGuard guard;
    int rc = hook_scene_params(guard, device, scene, p);
    if (rc != RT_OK) return rc;
    DevBuf<double> dr; DevBuf<uint32_t> drng; DevBuf<uint8_t> dc;
    HIP_TRY(dr.alloc((size_t) n * 6)); HIP_TRY(drng.alloc((size_t) n * 4)); HIP_TRY(dc.alloc((size_t) n * 3));
    HIP_TRY(dr.up(rays)); HIP_TRY(drng.up(rng_state));
    if (n) hipLaunchKernelGGL(k_trace_ray, dim3(blocks_for(n)), dim3(256), 0, 0, p, bounce_depth, n, dr.p, drng.p, dc.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(dc.down(colour_out)); HIP_TRY(drng.down(rng_state));
    return RT_OK;
}

int rt_dev_texture_colour_at(int32_t device, const rt_scene *scene, int32_t texture, int32_t n, const double *points, double *uv_out, uint8_t *colour_out) {
    if (!points || !colour_out || n < 0) return fail(RT_ERR_INVALID_ARGUMENT, "bad argument");
    if (!scene || texture < 0 || (size_t) texture >= scene->host.texRecs.size()) return fail(RT_ERR_INVALID_ARGUMENT, "texture index out of range");
    RenderParams p;
    DeviceGuard guard;
    int rc = hook_scene_params(guard, device, scene, p);
    if (rc != RT_OK) return rc;
    DevBuf<double> dp, duv; DevBuf<uint8_t> dc;
    HIP_TRY(dp.alloc((size_t) n * 3)); HIP_TRY(duv.alloc(uv_out ? (size_t) n * 2 : 0)); HIP_TRY(dc.alloc((size_t) n * 3));
    HIP_TRY(dp.up(points));
    if (n) hipLaunchKernelGGL(k_texture, dim3(blocks_for(n)), dim3(256), 0, 0, p, texture, n, dp.p, duv.p, dc.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    if (uv_out) HIP_TRY(duv.down(uv_out));
    HIP_TRY(dc.down(colour_out));
    return RT_OK;
}

} // extern "C"
