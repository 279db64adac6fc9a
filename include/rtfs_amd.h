/*
 * rtfs_amd.h -- C ABI of the MI355X-native per-pixel sampling path.
 *
 * This is the drop-in boundary for ONE hot path of Smaug123/ray-tracing-fsharp:
 *   Scene.render -> renderPixel -> traceOnce -> traceRay -> hitObject ->
 *   {BoundingBox.hits, Sphere.firstIntersection, InfinitePlane.intersection} -> Hittable.Reflection
 * (reference: RayTracing/Scene.fs:62-236).  The reference has no FFI of its own; every entry point
 * below names the reference function (file:line under /root/reference) whose role it takes, and
 * INTEGRATION.md shows the F# P/Invoke binding a maintainer would add.
 *
 * Conventions: POD only, caller-allocated outputs, int status (0 = RT_OK), no exceptions cross the
 * boundary, rt_last_error() is thread-local.  All geometry is IEEE double (Float.fs:82: `float` is
 * 64-bit); all colour is 8-bit (Pixel.fs:9-15); accumulators are int32 (Pixel.fs:78-85).
 *
 * The library is libamdhip64-only: no torch types appear here.  Device pointers are plain `void*`
 * so that any allocator (hipMalloc, torch.empty(..., device="cuda").data_ptr()) can own the memory.
 */
#ifndef RTFS_AMD_H
#define RTFS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 6 /* 5 (round 3): rt_scene_info.leaf_box_implied (was reserved), rt_dev_bbox_filter, rt_dev_pixel_candidates;
                          * 6: rt_scene_get_filter_tree */

/* ---- status codes ------------------------------------------------------------------------ */
enum {
    RT_OK = 0,
    RT_ERR_INVALID_ARGUMENT = 1, /* the reference would `failwith` / throw (e.g. Program.fs:69) */
    RT_ERR_NO_DEVICE = 2,        /* no HIP device visible: the product path never falls back to a CPU */
    RT_ERR_HIP = 3,              /* a HIP runtime call failed; rt_last_error() carries hipGetErrorString */
    RT_ERR_UNSUPPORTED = 4,      /* e.g. Texture.Arbitrary closures (Texture.fs:8,24) cannot cross a C ABI */
    RT_ERR_IO = 5
};

/* ---- Hittable (Hittable.fs:3-6) ------------------------------------------------------------ */
enum rt_hittable_kind {
    RT_HITTABLE_SPHERE = 0,           /* Hittable.Sphere: bounded, goes into the BoundingBoxTree */
    RT_HITTABLE_UNBOUNDED_SPHERE = 1, /* Hittable.UnboundedSphere: tested for every ray (Scene.fs:77-86) */
    RT_HITTABLE_INFINITE_PLANE = 2    /* Hittable.InfinitePlane: never in the tree (Hittable.fs:17-18) */
};

/* SphereStyle (Sphere.fs:10-37), in declaration order. */
enum rt_sphere_style {
    RT_SPHERE_LIGHT_SOURCE = 0,       /* LightSource of Texture */
    RT_SPHERE_LIGHT_SOURCE_CAP = 1,   /* LightSourceCap of Pixel */
    RT_SPHERE_PURE_REFLECTION = 2,    /* PureReflection of albedo * texture */
    RT_SPHERE_FUZZED_REFLECTION = 3,  /* FuzzedReflection of albedo * texture * fuzz * FloatProducer */
    RT_SPHERE_LAMBERT_REFLECTION = 4, /* LambertReflection of albedo * texture * FloatProducer */
    RT_SPHERE_DIELECTRIC = 5,         /* Dielectric of albedo * texture * ior * prob * FloatProducer */
    RT_SPHERE_GLASS = 6               /* Glass of albedo * texture * ior * FloatProducer */
};

/* InfinitePlaneStyle (InfinitePlane.fs:3-13), in declaration order. */
enum rt_plane_style {
    RT_PLANE_LIGHT_SOURCE = 0,
    RT_PLANE_PURE_REFLECTION = 1,
    RT_PLANE_LAMBERT_REFLECTION = 2,
    RT_PLANE_FUZZED_REFLECTION = 3
};

/*
 * One element of the `Hittable array` handed to Scene.make (Scene.fs:15).
 * Sphere.make style centre radius (Sphere.fs:325-337) / InfinitePlane.make style point normal
 * (InfinitePlane.fs:114-119).  FloatProducer arguments of the styles do not cross the boundary:
 * randomness comes from rt_render's `seed` (see DESIGN.md "Seeding").
 */
typedef struct rt_hittable {
    uint32_t kind;      /* rt_hittable_kind */
    uint32_t style;     /* rt_sphere_style or rt_plane_style, by kind */
    double   point[3];  /* sphere centre | a point on the plane */
    double   normal[3]; /* plane normal, already unitised by the caller (InfinitePlane.make takes a UnitVector); spheres: ignored */
    double   radius;    /* spheres; may be negative (Sphere.fs:321 "flipped") */
    double   albedo;    /* float<albedo>, must lie in [0,1] (Pixel.fs:143) */
    double   fuzz;      /* FuzzedReflection */
    double   ior;       /* Dielectric / Glass boundaryRefractance */
    double   prob;      /* Dielectric refraction probability */
    uint8_t  rgb[3];    /* Texture.Colour / LightSourceCap colour / plane colour */
    uint8_t  reserved;
    int32_t  texture;   /* -1: Texture.Colour rgb.  >=0: index into the rt_texture array (sphere styles with a texture only) */
} rt_hittable;

/* ---- Textures (Texture.fs:6-72): closures are enumerated ----------------------------------- */
enum rt_texture_kind {
    RT_TEXTURE_COLOUR = 0,    /* ParameterisedTexture.Colour */
    RT_TEXTURE_CHECKERED = 1, /* ParameterisedTexture.Checkered (even, odd, gridSize), Texture.fs:56-62 */
    RT_TEXTURE_IMAGE = 2,     /* ParameterisedTexture.Image rows (Texture.fs:63-67) */
    RT_TEXTURE_UV_RAMP = 3    /* the two ParameterisedTexture.Arbitrary closures of SampleImages.fs:606-627 */
};

enum rt_ramp_source { RT_RAMP_CONST = 0, RT_RAMP_U = 1, RT_RAMP_V = 2 };
enum rt_walk_tree { RT_WALK_TREE_SAH = 0, RT_WALK_TREE_REFERENCE = 1, RT_WALK_TREE_TUNED = 2 /* reported after rt_scene_tune; not a creation option */ };

typedef struct rt_texture {
    uint32_t kind;          /* rt_texture_kind */
    uint8_t  rgb[3];        /* COLOUR; UV_RAMP constants */
    uint8_t  ramp_src[3];   /* UV_RAMP: per channel rt_ramp_source; U -> byte(u*255.0), V -> byte(v*255.0) (truncating) */
    uint8_t  reserved[2];
    int32_t  even, odd;     /* CHECKERED: indices into the same texture array (must be < own index) */
    double   grid_size;     /* CHECKERED */
    int32_t  width, height; /* IMAGE */
    const uint8_t *texels;  /* IMAGE: height*width*3 bytes, laid out as ParameterisedTexture.Image img.[y].[x]
                               i.e. AFTER ofImage's row reversal (Texture.fs:34); copied by rt_scene_create */
    double   map_centre[3]; /* interpret = Sphere.planeMapInverse map_radius map_centre (Sphere.fs:55-61), */
    double   map_radius;    /*   read from the texture a hittable points at (the root of a Checkered tree)  */
} rt_texture;

/* ---- Camera (Camera.fs:3-28) ----------------------------------------------------------------- */
typedef struct rt_camera {
    double  view_origin[3], view_dir[3];   /* View : Ray */
    double  xaxis_origin[3], xaxis_dir[3]; /* ViewportXAxis : Ray */
    double  yaxis_origin[3], yaxis_dir[3]; /* ViewportYAxis : Ray (only its direction is used, Scene.fs:140) */
    double  viewport_width, viewport_height, focal_length;
    int32_t samples_per_pixel;
    int32_t bounce_depth;                  /* Camera.fs:58 hard-codes 150; callers may override */
} rt_camera;

/* Camera.makeBasic samplesPerPixel focalLength aspectRatio origin viewDirection viewUp (Camera.fs:34-59;
 * Plane.makeNormalTo' Plane.fs:22-38; Plane.basis Plane.fs:82-97).  view_direction must be unit length. */
int rt_camera_make_basic(int32_t samples_per_pixel, double focal_length, double aspect_ratio,
                         const double origin[3], const double view_direction[3], const double view_up[3],
                         rt_camera *out);

/* ---- Scene (Scene.fs:5-28, BoundingBoxTree.fs:9-43) ------------------------------------------ */
typedef struct rt_scene rt_scene;

typedef struct rt_scene_info {
    int32_t n_bounded;      /* Hittable.Sphere count (leaves of the tree) */
    int32_t n_unbounded;    /* UnboundedSphere + InfinitePlane count, original order kept (Array.partition) */
    int32_t n_nodes;        /* BoundingBoxTree nodes, leaves included (2*n_bounded-1): the size of rt_scene_get_tree's arrays */
    int32_t tree_depth;     /* of BoundingBoxTree.make's tree (what rt_scene_get_tree reports) */
    int32_t n_textures;
    int32_t lds_resident;   /* 1 if the flattened scene fits the 160 KiB LDS image and the LDS kernel is used */
    int32_t walk_tree;      /* RT_WALK_TREE_*: the tree the device image holds */
    int32_t walk_tree_depth;
    int64_t scene_bytes;    /* bytes of the flattened device image (without texels) */
    int64_t texel_bytes;
    int32_t walk_tree_nodes; /* nodes of the tree the device image holds: the size of rt_scene_get_walk_tree's arrays (= n_nodes
                                until rt_scene_tune thins the tree) */
    int32_t leaf_box_implied; /* 1: the scene meets the bounds under which the timed kernel's leaf pass need not evaluate a Leaf's own
                                 BoundingBox.hits when Sphere.firstIntersection has found a hit (csrc/rt_device.h,
                                 leaf_test_object_exact: every bounded radius in (0, 100], coordinates within 1000, r_max * extent <= 500);
                                 0: the leaf pass evaluates it for every candidate.  Diagnostic; results never depend on it. */
} rt_scene_info;

/* Scene.make (Scene.fs:15-28): partitions bounded/unbounded, builds the BoundingBoxTree on the host,
 * builds the tree the device walks over the same Leaf boxes (rt_set_walk_tree), flattens that (DFS pre-order, on-hit and
 * on-miss successor per node) and keeps a host copy; device copies are made lazily per device. */
int rt_scene_create(const rt_hittable *hittables, size_t n_hittables,
                    const rt_texture *textures, size_t n_textures, rt_scene **out);
/* Per-scene settings, handed over at creation instead of through the process-wide rt_set_walk_tree default. */
typedef struct rt_scene_options {
    uint32_t struct_size; /* sizeof(rt_scene_options) as the caller compiled it (fields beyond it keep their defaults) */
    int32_t  walk_tree;   /* RT_WALK_TREE_*; -1 = the process default (rt_set_walk_tree) */
} rt_scene_options;
/* rt_scene_create with explicit options (NULL = all defaults).  Thread-safe: reads no process-wide setting when every
 * option is given. */
int rt_scene_create_ex(const rt_hittable *hittables, size_t n_hittables, const rt_texture *textures, size_t n_textures,
                       const rt_scene_options *options, rt_scene **out);
void rt_scene_destroy(rt_scene *scene);
int rt_scene_get_info(const rt_scene *scene, rt_scene_info *out);
/* Flattened tree for inspection/tests: skip[n_nodes], prim[n_nodes] (-1 for Branch), boxes[n_nodes*6] as
 * (minx,maxx,miny,maxy,minz,maxz).  Any pointer may be NULL. */
int rt_scene_get_tree(const rt_scene *scene, int32_t *skip, int32_t *prim, double *boxes);
/* The same arrays for the tree the device image holds (rt_set_walk_tree): identical to rt_scene_get_tree's under
 * RT_WALK_TREE_REFERENCE; under RT_WALK_TREE_SAH another binary tree over the same Leaf boxes, every Branch box again the
 * exact union of the Leaf boxes below it. */
int rt_scene_get_walk_tree(const rt_scene *scene, int32_t *skip, int32_t *prim, double *boxes);
/* The same tree as the TIMED kernel walks it (csrc/rt_device.h, "the node loop of the timed variant"): walk_tree_nodes records in the
 * order the device image stores them -- by depth, so that the top of the tree is a prefix (what a scene too large for the LDS keeps
 * there) -- with explicit links.  boxes[p*6 ..] = lo, hi per axis in single precision, rounded outward from the walk tree's boxes;
 * links[p*5 ..] = { record visited after a hit, record visited after a miss, queue entry of a Leaf (0 for a Branch: 0x4000 | index
 * in the image's object order, or 0x80000000 | index from 16384 objects), shift that pushes it, the Leaf's hittable as an index
 * into rt_scene_create's input (-1 for a Branch) }, records counted from 0, walk_tree_nodes = "tree exhausted".  A Leaf's two links are equal (its exact tests are queued, the walk goes on).  Host-side
 * only (no device needed): for tests of the layout; the format is this build's, not part of the boundary's contract. */
int rt_scene_get_filter_tree(const rt_scene *scene, float *boxes, int32_t *links);

/* ---- Tuning the walk tree to a camera (no counterpart in the reference: BoundingBoxTree.make knows no rays) ----------
 * Renders a PROBE with the scene as it stands -- 16 image rows spread over the frame, through the counting kernel, which logs a
 * thinned-out sample of the rays it traces (8192 are used: origin and direction) -- and rebuilds the tree the device walks
 * from them: split costs are the number of probe rays that hit a candidate box (not its area), and Branch boxes that nearly
 * every arriving ray hits are not tested at all (their children take their place; csrc/rt_scene.h "thinning").  Every pixel
 * stays the same bit for bit, as under rt_set_walk_tree and for the same reason; what changes is aabb_tests (final scene:
 * 23.8 -> about 17 per ray) and the frame time.  rt_scene_get_walk_tree then reports an n-ary tree in the same pre-order/skip
 * form, rt_scene_get_info.walk_tree = RT_WALK_TREE_TUNED, walk_tree_nodes shrinks (n_nodes stays the size of BoundingBoxTree.make's
 * own tree; size rt_scene_get_walk_tree's arrays by walk_tree_nodes).  The probe is deterministic (which rays are logged depends
 * on their random streams only, and the log is sorted), so the same call yields the same tree; a probe whose log still overflows
 * after being thinned to one ray in 2^20 would hold a scheduling-dependent subset, and such a scene is left untuned (tuned = 0).
 * The replacement is failure-atomic: host tree and device copies change together or not at all.
 * A scene that walks the reference's own tree (RT_WALK_TREE_REFERENCE, or fewer than 3 bounded spheres, or non-finite boxes)
 * is left alone: tuned = 0.  Must not run concurrently with renders of the same scene: it replaces the device images (after
 * waiting for the devices that hold one).  Typical use: once after rt_scene_create, with the camera and image size of the
 * frames to come.  Cost on the final scene (round 3): ~1.3 ms of GPU time for the probe (16 rows at a reduced sample count, the
 * timed kernel variant) + ~4 ms of host time for the build (subtrees on separate threads), ~6.5 ms wall in all, against ~13 ms
 * saved per 2401x1601x500spp frame: it pays on the FIRST frame of such a scene (123 ms against 130 ms untuned, 116.6 ms from the
 * second frame on; bench.py reports all three). */
typedef struct rt_tune_info {
    uint32_t struct_size;      /* sizeof(rt_tune_info) as the caller compiled it */
    int32_t  tuned;            /* 1: the walk tree was replaced */
    int32_t  probe_rows;       /* image rows rendered for the probe */
    int32_t  probe_rays;       /* rays the build used */
    int32_t  nodes_before, nodes_after;
    double   box_tests_before, box_tests_after; /* BoundingBox.hits calls per probe ray, counted on the host */
    double   probe_ms, build_ms;
} rt_tune_info;
int rt_scene_tune(rt_scene *scene, const rt_camera *camera, int32_t max_width_coord, int32_t max_height_coord, uint64_t seed,
                  int32_t device, rt_tune_info *info /* may be NULL */);
/* The host half alone, with the caller's own probe: rays[n_rays][6] = origin xyz, direction xyz (at most 8192 of them are used,
 * evenly spaced).  Touches a GPU only to replace device copies of the image that already exist.  probe_rows and probe_ms stay 0. */
int rt_scene_tune_rays(rt_scene *scene, const double *rays, size_t n_rays, rt_tune_info *info /* may be NULL */);

/* ---- Render (Scene.render, Scene.fs:196-236) ---------------------------------------------------- */
typedef struct rt_stats {
    uint64_t rays;        /* Scene.hitObject calls (Scene.fs:62) = primary + secondary rays */
    uint64_t aabb_tests;  /* BoundingBox.hits calls (BoundingBox.fs:30) */
    uint64_t prim_tests;  /* Hittable.hits calls (Hittable.fs:27): sphere + plane tests, unbounded included */
    uint64_t reflections; /* Hittable.Reflection calls (Hittable.fs:8-12) = path vertices shaded */
    uint64_t samples;     /* Scene.traceOnce calls (Scene.fs:118) */
    uint64_t pixels;      /* pixels rendered by this call */
    uint64_t pixels_early;/* pixels that stopped after 2*firstTrial+1 samples (Scene.fs:185-188) */
    double   kernel_ms;   /* device time of the render kernel, HIP events on the launch stream */
    double   total_ms;    /* wall time of the call, host side */
} rt_stats;

#define RT_RENDER_COUNTERS 1u /* fill rays/aabb_tests/prim_tests/reflections (slightly slower kernel variant) */

/*
 * Image geometry (Scene.fs:208-209,219,226): rows = 2*max_height_coord+1, cols = 2*max_width_coord+1;
 * image row index r (0 = top) maps to row = max_height_coord - r - 1, column index c to col = c - max_width_coord.
 *
 * A call renders the image rows r = row_first + i*row_stride for i in [0, n_rows): (0, 1, rows) is the whole
 * frame; (rank, world, ceil((rows-rank)/world)) is one rank's interleaved shard.  RNG streams are keyed by
 * (seed, global pixel index r*cols+c, sample index), so a shard's output does not depend on the sharding.
 *
 * accum: n_rows*cols*4 int32 = PixelStats {Count; SumRed; SumGreen; SumBlue} (Pixel.fs:78-85).
 * rgb:   n_rows*cols*3 uint8 = PixelStats.mean (Pixel.fs:103-108); may be NULL.
 */
int rt_render(const rt_scene *scene, const rt_camera *camera,
              int32_t max_width_coord, int32_t max_height_coord, uint64_t seed,
              int32_t device, int32_t row_first, int32_t row_stride, int32_t n_rows,
              uint32_t flags, int32_t *accum_host, uint8_t *rgb_host, rt_stats *stats);

/* Same, with outputs left in device memory (d_accum / d_rgb are device pointers on `device`) and the launch
 * enqueued on `stream` (a hipStream_t, NULL = the null stream).  If `stats` is non-NULL the call synchronises
 * the stream and fills it; with stats == NULL it returns right after the launch.  Any number of launches may be in flight:
 * each takes its counters, work queue and camera from stream-ordered scratch of its own.  The caller's current HIP device
 * is left as it was. */
int rt_render_device(const rt_scene *scene, const rt_camera *camera,
                     int32_t max_width_coord, int32_t max_height_coord, uint64_t seed,
                     int32_t device, int32_t row_first, int32_t row_stride, int32_t n_rows,
                     uint32_t flags, void *d_accum, void *d_rgb, void *stream, rt_stats *stats);

/* Launch settings of one render call.  0 in any field = that field's process default (the rt_set_* calls below, which
 * exist for bench sweeps); a call that passes a fully specified struct reads no process-wide state.  No setting ever
 * changes a result -- only which wave traces which sample when. */
typedef struct rt_render_options {
    uint32_t struct_size;   /* sizeof(rt_render_options) as the caller compiled it */
    int32_t  block_threads; /* 256, 512, 768 or 1024 threads per workgroup */
    int32_t  chunk_pixels;  /* pixels per wave work unit, <= 64 */
    int32_t  blocks_per_cu; /* cap on resident workgroups per CU */
    int32_t  yield_lanes;   /* lane-scheduling thresholds (DESIGN.md "Kernel") */
    int32_t  refill_lanes;
    int32_t  passes;        /* 1 fused kernel, 2 two passes (phase 1 + decision, cost-ordered phase 2); 0 = default (by shard size) */
    int32_t  park_lanes;    /* capacity of a wave's pool of parked rare-style paths (<= 256); 0 = default, -1 = never park */
} rt_render_options;
int rt_render_device_ex(const rt_scene *scene, const rt_camera *camera,
                        int32_t max_width_coord, int32_t max_height_coord, uint64_t seed,
                        int32_t device, int32_t row_first, int32_t row_stride, int32_t n_rows,
                        uint32_t flags, void *d_accum, void *d_rgb, void *stream,
                        const rt_render_options *options, rt_stats *stats);

/*
 * Scene.render for a whole frame on SEVERAL GPUs of one node from ONE process (Scene.fs:196-236):
 * device i of `devices[0..n_devices)` renders the image rows r = i, i + n_devices, ... (interleaved: adaptive sampling makes
 * sky rows ~11/spp the cost of object rows) on a stream of its own, all devices concurrently; then ONE gather brings the
 * per-device PixelStats buffers together and the frame is handed back de-interleaved in host memory:
 * accum_host rows*cols*4 int32, rgb_host rows*cols*3 uint8 (may be NULL).  Streams are keyed by the global pixel index, so
 * the frame is bit-identical for any device count.
 *
 * gather: RT_GATHER_RCCL   ncclGroupStart / ncclRecv x (n-1) on devices[0] / ncclSend on the others / ncclGroupEnd over
 *                          xGMI (librccl.so is loaded on first use), then one strided device-to-host copy from devices[0];
 *         RT_GATHER_PEER   hipMemcpyPeerAsync to devices[0] instead of RCCL (also works when `devices` repeats an id);
 *         RT_GATHER_HOST   every device copies its own shard to the host over its own PCIe link (no device-side gather);
 *         RT_GATHER_AUTO   RCCL when n_devices > 1, the ids are distinct and librccl.so loads; else PEER.
 * stats (may be NULL): n_devices entries, one per device's shard; kernel_ms is that device's render time, total_ms the
 * wall time of the whole call (the same in every entry).
 */
enum rt_gather { RT_GATHER_AUTO = 0, RT_GATHER_RCCL = 1, RT_GATHER_PEER = 2, RT_GATHER_HOST = 3 };
int rt_render_frame(const rt_scene *scene, const rt_camera *camera,
                    int32_t max_width_coord, int32_t max_height_coord, uint64_t seed,
                    const int32_t *devices, int32_t n_devices, uint32_t flags, int32_t gather,
                    const rt_render_options *options,
                    int32_t *accum_host, uint8_t *rgb_host, rt_stats *stats);

/* ---- Output side (ImageOutput.fs:11-30,163-197) -------------------------------------------------- */
uint8_t rt_gamma_correct(uint8_t b); /* PixelOutput.correct (ImageOutput.fs:11-18) */
/* ImageOutput.writePpm gammaCorrect pixels file (ImageOutput.fs:163-197): P3, no trailing newline. */
int rt_write_ppm(const char *path, const uint8_t *rgb, int32_t rows, int32_t cols, int32_t gamma_correct);
/* Same bytes into a caller buffer; returns the length needed (excluding NUL) or a negative status. */
int64_t rt_format_ppm(const uint8_t *rgb, int32_t rows, int32_t cols, int32_t gamma_correct,
                      char *out, size_t out_capacity);

/* ImageOutput.resume's temp-file bytes (ImageOutput.fs:131-161): per pixel `<row>,<col>\n` in ASCII (0 is written as NO
 * digits, ImageOutput.fs:115-129) followed by the three raw colour bytes.  Returns the length, or a negative status. */
int64_t rt_format_pixel_map(const uint8_t *rgb, int32_t rows, int32_t cols, uint8_t *out, size_t out_capacity);
/* ImageOutput.readPixelMap (ImageOutput.fs:46-113): fills rgb_out for every pixel the data names and present_out[r*cols+c]=1
 * (may be NULL); a truncated tail is ignored as in the reference.  Returns the pixel count, or a negative status. */
int64_t rt_parse_pixel_map(const uint8_t *data, size_t n, int32_t rows, int32_t cols, uint8_t *rgb_out, uint8_t *present_out);

/* ---- Runtime ------------------------------------------------------------------------------------- */
int rt_device_count(void);         /* 0 when no HIP device is visible (never an error) */
const char *rt_last_error(void);   /* thread-local message of the last failing call */
int rt_abi_version(void);
/* sizeof of the ABI structs as compiled: 0 rt_hittable, 1 rt_texture, 2 rt_camera, 3 rt_scene_info, 4 rt_stats,
 * 5 rt_render_options, 6 rt_scene_options, 7 rt_tune_info (bindings check their mirrors). */
size_t rt_abi_sizeof(int which);
/* Byte offset of field number `field` (declaration order, from 0) of struct `which` (numbering of rt_abi_sizeof), or
 * (size_t)-1 past the last field: a binding in another language asserts its own layout against these at start-up
 * (the F# StructLayout(Sequential) mirrors; ray-tracing-fsharp_amd/_lib.py: the ctypes ones). */
size_t rt_abi_offsetof(int which, int field);
/* Tunables of the render kernel: threads per workgroup (256, 512, 768 or 1024) and pixels per wave work unit (<= 64).
 * 0 keeps the default.  Process-wide DEFAULTS for calls that pass no rt_render_options; meant for bench sweeps. */
int rt_set_launch_config(int32_t block_threads, int32_t chunk_pixels, int32_t blocks_per_cu);
/* Lane-scheduling thresholds of the render kernel (DESIGN.md "Kernel"): a stage yields once `yield_lanes` lanes wait for
 * another stage; idle lanes are refilled once `refill_lanes` are idle.  0 keeps the default.  Results never depend on them. */
int rt_set_schedule(int32_t yield_lanes, int32_t refill_lanes);
/* Which binary tree over the Leaf boxes the device walks, for scenes created AFTERWARDS.  RT_WALK_TREE_SAH (default): a
 * surface-area-heuristic build, ~14 % fewer box tests per ray on the reference's scenes; RT_WALK_TREE_REFERENCE:
 * BoundingBoxTree.make's own tree (BoundingBoxTree.fs:9-43) with Array.sortBy taken as a STABLE sort, i.e. the oracle's
 * tree: the box-test count then equals the oracle's.  (.NET's Array.sortBy is an unstable introsort and every small sphere
 * of the final scene ties on Min.y, so the real reference's tree -- and its count -- may differ from both.)  The hit a ray
 * returns -- hence every pixel -- is the same bit for bit under every such tree (rt_scene.h "the tree the device WALKS");
 * only the aabb_tests statistic differs. */
int rt_set_walk_tree(int32_t kind);
/* 1: always the fused kernel; 2: always two passes (phase 1 + decision, cost-ordered phase 2); 0: choose by shard size.
 * Results never depend on it. */
int rt_set_passes(int32_t passes);
/* Capacity of a wave's pool of parked paths: a path whose hit is neither an untextured light source nor an untextured
 * Lambert sphere is set aside (88 bytes of state, in global memory) and shaded later together with others of its kind
 * (DESIGN.md "Kernel").  0 = default (64), -1 = never park (such paths are shaded in their lane).  Results never depend on it. */
int rt_set_park(int32_t park_lanes);
/* Diagnostic: wave-level stage executions of the calling THREAD's last render that asked for stats with
 * RT_RENDER_COUNTERS set: {refill stages, node-loop trips, leaf stages, shade stages, lanes refilled, lanes shaded, sum of
 * wave lifetimes and first-start-to-last-end span (both in 100 MHz ticks), waves launched, general-reflection stages, lanes in them,
 * lanes parked, shader-clock cycles summed over the waves inside the refill / general-reflection / walk / shade stages}. */
int rt_last_stage_stats(uint64_t out[16]);

/*
 * ---- Device unit hooks ---------------------------------------------------------------------------
 * Run ONE device function of the path over n inputs on the GPU, so that the reference's unit tests
 * (the .fs files under RayTracing.Test) can be replayed against the very code the render kernel inlines.
 * All pointers are HOST pointers; the hooks copy in, launch, copy out.
 */
/* FloatProducer (Float.fs:14-76): from state[4] produce n doubles with Get(). */
int rt_dev_float_producer(int32_t device, const uint32_t state[4], int32_t n, double *out);
/* Stream seeding (DESIGN.md "Seeding"): state for (seed, pixel, sample). */
int rt_dev_stream_state(int32_t device, uint64_t seed, int32_t n, const uint64_t *pixel, const uint32_t *sample,
                        uint32_t *state_out /* n*4 */);
/* BoundingBox.inverseDirections + hits (BoundingBox.fs:25-94). rays: n*6 (origin, unit dir); boxes: n*6 (min xyz, max xyz). */
int rt_dev_bbox_hits(int32_t device, int32_t n, const double *rays, const double *boxes, int32_t *hit_out);
/* The timed kernel's node loop does not run BoundingBox.hits on the boxes ABOVE the leaves: it runs a conservative
 * single-precision filter F with hits(box) => F(box) (csrc/rt_device.h, "the node loop of the timed variant"), and the leaf pass
 * then makes the Leaf's own BoundingBox.hits exactly -- the set of spheres tested is the reference's (Scene.fs:39-60).  This hook
 * runs both on n (ray, box) pairs: out[i] bit 0 = the exact test, bit 1 = F in the loop's own instruction forms, bit 2 = F as
 * compiled C++.  Boxes are rounded outward to single precision as the scene image does; bmax (>= 0) enlarges the margin's scale
 * beyond the batch's largest |coordinate| (the scene image uses the largest |coordinate| of its tree). */
int rt_dev_bbox_filter(int32_t device, int32_t n, const double *rays, const double *boxes, double bmax, int32_t *out);
/* The timed kernel walks the tree ONCE per pixel for all of the pixel's camera rays (Scene.traceOnce, Scene.fs:129-144): with the
 * pyramid from the eye through the pixel's patch of the viewport it collects the Leaves any of those rays can reach (at most four;
 * csrc/rt_device.h, pixel_candidates), and a camera ray then starts with those Leaves queued for their exact tests instead of
 * walking.  This hook returns that set for n pixels given as (row, col) pairs in the reference's coordinates (row = maxH - r - 1,
 * col = c - maxW, Scene.fs:219,226): leaves_out[i*4 .. i*4+3] = hittable indices (input order of rt_scene_create), -1 padded;
 * leaves_out[i*4] = -2 when the pixel's camera rays walk the tree as all other rays do (more than four Leaves in reach). */
int rt_dev_pixel_candidates(int32_t device, const rt_scene *scene, const rt_camera *camera, int32_t max_width_coord, int32_t max_height_coord,
                            int32_t n, const int32_t *row_col, int32_t *leaves_out);
/* Sphere.firstIntersection (Sphere.fs:349-386). spheres: n*4 (centre xyz, radius). t_out = NaN when ValueNone. */
int rt_dev_sphere_first_intersection(int32_t device, int32_t n, const double *rays, const double *spheres, double *t_out);
/* InfinitePlane.intersection (InfinitePlane.fs:125-136). planes: n*6 (point, unit normal). */
int rt_dev_plane_intersection(int32_t device, int32_t n, const double *rays, const double *planes, double *t_out);
/* Pixel.combine / Pixel.darken (Pixel.fs:136-151). a,b: n*3 bytes; albedo: n. */
int rt_dev_pixel_combine(int32_t device, int32_t n, const uint8_t *a, const uint8_t *b, uint8_t *out);
int rt_dev_pixel_darken(int32_t device, int32_t n, const uint8_t *p, const double *albedo, uint8_t *out);
/* Hittable.Reflection (Hittable.fs:8-12 -> Sphere.fs:150-300 | InfinitePlane.fs:43-99) for hittable `index` of the
 * scene (index into the array given to rt_scene_create).  Per item: ray_in n*6, colour_in n*3, strike n*3,
 * rng_state n*4 (in/out).  Outputs: absorbed[n] (1 = ValueSome colour), colour_out n*3, ray_out n*6. */
int rt_dev_reflection(int32_t device, const rt_scene *scene, int32_t n, const int32_t *index,
                      const double *ray_in, const uint8_t *colour_in, const double *strike,
                      uint32_t *rng_state, int32_t *absorbed, uint8_t *colour_out, double *ray_out);
/* Scene.hitObject (Scene.fs:62-91): hit_index = index into the rt_scene_create array or -1; strike n*3; counters optional (n*2: aabb, prim). */
int rt_dev_hit_object(int32_t device, const rt_scene *scene, int32_t n, const double *rays,
                      int32_t *hit_index, double *strike, uint32_t *counters);
/* The same through the render kernel's timed route: scene staged into LDS, the hand-written node loop with its per-lane queue of
 * pending leaf tests, leaf passes between its runs, the unbounded objects last (RT_ERR_UNSUPPORTED when the scene does not fit the LDS). */
int rt_dev_hit_object_lds(int32_t device, const rt_scene *scene, int32_t n, const double *rays, int32_t *hit_index, double *strike);
/* Scene.traceRay (Scene.fs:93-114) from White for given rays and RNG states. colour_out n*3. */
int rt_dev_trace_ray(int32_t device, const rt_scene *scene, int32_t bounce_depth, int32_t n, const double *rays,
                     uint32_t *rng_state, uint8_t *colour_out);
/* Texture lookup incl. Sphere.planeMapInverse (Sphere.fs:55-61, Texture.fs:50-67): uv_out n*2, colour_out n*3. */
int rt_dev_texture_colour_at(int32_t device, const rt_scene *scene, int32_t texture, int32_t n, const double *points,
                             double *uv_out, uint8_t *colour_out);
/* IEEE-754 conformance probes of the device arithmetic the path relies on: op 0: 1.0/x, 1: sqrt(x), 2: rint(x),
 * 3: x/y, 4: pow5(x) = Math.Pow(x, 5.0) (Sphere.fs:290), 5: the path's sqrt for operands > 1e-8, 6: its 1.0/sqrt(x) for
 * operands >= 1e-8 (both must equal the correctly rounded results), 7: Math.Acos(x), 8: Math.Sin(x), 9: Math.Atan2(x, y) as the
 * texture maps use them (Sphere.fs:59-60, Texture.fs:58): the correctly rounded values (csrc/rt_trig.h).
 * a,b: n doubles (b may be NULL for unary ops). */
int rt_dev_arith(int32_t device, int32_t op, int32_t n, const double *a, const double *b, double *out);

#ifdef __cplusplus
}
#endif
#endif /* RTFS_AMD_H */
