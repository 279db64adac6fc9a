// rt_device.h -- device functions of the per-pixel sampling path, written for gfx950 (CDNA4, wave64).
//
// Each function names the reference F# it takes the role of (paths relative to /root/reference/RayTracing).
// Arithmetic contract (DESIGN.md "Exactness"): IEEE double throughout, no contraction (the file is built with
// -ffp-contract=off and carries the pragma below), no fmin/fmax (NaN ordering of BoundingBox.fs:57-58 is kept by
// writing the compares as the reference does), 1.0/x and sqrt are the correctly rounded device sequences,
// Math.Round = v_rndne_f64 (rint).  The structure is NOT the reference's: no heap Ray objects, no recursion,
// a stackless skip-link walk over a pre-order tree image, byte colour packed in one VGPR, material handling
// split into decide/refract/reflect/fuzz stages so that lanes with different SphereStyles share instructions.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "rt_trig.h"

#pragma clang fp contract(off)

namespace rtd {

#define RTD_INLINE __device__ __forceinline__
#define RTD_AS3 __attribute__((address_space(3)))

typedef double d2 __attribute__((ext_vector_type(2)));
typedef int i2 __attribute__((ext_vector_type(2)));
typedef int i4 __attribute__((ext_vector_type(4)));

// ---- Float.fs:82-96 ---------------------------------------------------------------------------------
#define RTD_TOL 0.00000001
RTD_INLINE bool feq(double a, double b) { return fabs(a - b) < RTD_TOL; }
RTD_INLINE bool fpos(double a) { return a > RTD_TOL; }
enum { CMP_GT = 0, CMP_EQ = 1, CMP_LT = 2 };
RTD_INLINE int fcmp(double a, double b) {
    if (fabs(a - b) < RTD_TOL) return CMP_EQ;
    return (a < b) ? CMP_LT : CMP_GT;
}

// ---- Point.fs:18-45 ------------------------------------------------------------------------------------
struct V3 { double x, y, z; };
RTD_INLINE V3 mk(double x, double y, double z) { V3 v; v.x = x; v.y = y; v.z = z; return v; }
RTD_INLINE double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RTD_INLINE V3 vsub(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
RTD_INLINE V3 vscale(double s, V3 v) { return mk(s * v.x, s * v.y, s * v.z); }
// Ray.walkAlongRay (Ray.fs:42-43): o + (v * m) per component
RTD_INLINE V3 walk(V3 o, V3 v, double m) { return mk(o.x + (v.x * m), o.y + (v.y * m), o.z + (v.z * m)); }
// Correctly rounded sqrt and reciprocal for operands in the NORMAL range.  The compiler's expansions of sqrt(x) and 1.0/x
// (v_rsq_f64 / v_rcp_f64 + Newton steps in fma, then one exactly-computed-residual correction -- the sequence that makes the
// result the correctly rounded one) spend 5 of 18 and 4 of 11 instructions on rescaling denormal / huge operands and on the
// zero/inf fix-ups.  Every caller below passes a value that has already compared > 1e-8 (or is NaN, which propagates), so the
// rescaling never triggers and the same arithmetic without it returns the same bits; +inf is restored by the caller's one
// class test.  tests/test_gpu_parity.py::test_normal_range_sqrt_and_reciprocal checks both against the host's sqrt and
// division on 40 M operands over [1e-8, 1e300].
RTD_INLINE double sqrt_core(double x) { // x in [2^-767, 2^1023]: LLVM's f64 sqrt lowering without its input scaling
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = y * 0.5;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    double e = fma(-g, g, x);
    g = fma(e, h, g);
    e = fma(-g, g, x);
    return fma(e, h, g);
}
RTD_INLINE double rcp_core(double s) { // 1.0 / s for s in [2^-511, 2^511]: LLVM's f64 division lowering without scale and fix-up
    double r = __builtin_amdgcn_rcp(s);
    double e = fma(-s, r, 1.0);
    r = fma(r, e, r);
    e = fma(-s, r, 1.0);
    r = fma(r, e, r);
    e = fma(-s, r, 1.0); // numerator 1.0: the quotient estimate IS r; residual computed exactly, one last correction
    return fma(e, r, r);
}
RTD_INLINE double sqrt_above_tol(double x) { // sqrt(x) for x > 1e-8 or NaN
    const double g = sqrt_core(x);
    return __builtin_isinf(x) ? x : g;
}
RTD_INLINE double inv_sqrt_above_tol(double x) { // 1.0 / sqrt(x), both roundings as written, for |x| >= 1e-8 or NaN
    const double q = rcp_core(sqrt_core(x));
    return __builtin_isinf(x) ? 0.0 : q; // sqrt(+inf) = +inf, 1.0 / +inf = +0.0
}
// Vector.unitise (Point.fs:28-35) == Ray.make' / Ray.overwriteWithMake's direction part (Ray.fs:11-34)
RTD_INLINE bool unitise(V3 v, V3 &out) {
    double d = dot(v, v);
    if (feq(d, 0.0)) return false;
    double factor = inv_sqrt_above_tol(d); // 1.0 / sqrt d, d >= 1e-8 here
    out = vscale(factor, v);
    return true;
}

// ---- FloatProducer (Float.fs:14-76) ------------------------------------------------------------------------
struct Rng { uint32_t x, y, z, w; };
RTD_INLINE uint32_t rng_next(Rng &r) { // generateInt32, Float.fs:14-20
    uint32_t t = r.x ^ (r.x << 11);
    r.x = r.y;
    r.y = r.z;
    r.z = r.w;
    r.w = r.w ^ (r.w >> 19) ^ (t ^ (t >> 8));
    return r.w;
}
// toInt is a byte reversal (Float.fs:22-27); toDouble divides by UInt32.MaxValue (Float.fs:29).
// The division is replaced by an identity: for every uint32 u, u / 4294967295.0 == fma(u, 2^-64 + 2^-96, u * 2^-32)
// bit for bit (u/(2^32-1) = u*2^-32*(1 + 2^-32 + 2^-64 + ...); the fma carries the first three terms exactly and rounds
// once, and the remaining tail can never reach a rounding boundary).  Checked EXHAUSTIVELY over all 2^32 inputs by
// tests/test_rng_division_identity.py; the oracle keeps the literal division.
RTD_INLINE double rng_get(Rng &r) {
    const double x = (double) __builtin_bswap32(rng_next(r));
    return fma(x, 0x1p-64 + 0x1p-96, x * 0x1p-32);
}

// Stream seeding (DESIGN.md "Seeding"): every pixel owns a SplitMix64 sequence keyed by (seed, global pixel index); sample
// s of the pixel takes outputs 2s+1 and 2s+2 of it as the four xorshift128 words.
__host__ __device__ __forceinline__ uint64_t mix64(uint64_t z) { // SplitMix64 finaliser; integer-only, also used by the host
    z ^= z >> 30;
    z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27;
    z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}
#define RTD_GOLDEN 0x9E3779B97F4A7C15ull
// x % (2^31 - 1) for a 32-bit x: 2^31 = 1 (mod 2^31 - 1), so x = hi * 2^31 + lo is congruent to hi + lo <= 2^31, one conditional
// subtraction away from the residue (the compiler's magic-number sequence uses two quarter-rate multiplies instead).
RTD_INLINE uint32_t mod_m31(uint32_t x) {
    const uint32_t r = (x & 0x7FFFFFFFu) + (x >> 31);
    return r >= 2147483647u ? r - 2147483647u : r;
}
// item / per with wave-uniform `per` < 2^23 and item < 2^22 (callers check): a float multiply by rcp = 1.0f / per lands within one
// of the quotient (relative error < 2^-22), one exact remainder fixes it; ~9 full-rate instructions instead of the ~23 of the
// integer-division expansion.
RTD_INLINE uint32_t div_uniform(uint32_t item, uint32_t per, float rcp) {
    uint32_t j = (uint32_t) ((float) item * rcp);
    const int32_t r = (int32_t) item - (int32_t) __umul24(j, per);
    if (r < 0) j -= 1u;
    else if ((uint32_t) r >= per) j += 1u;
    return j;
}
RTD_INLINE uint64_t pixel_key(uint64_t seedKey, uint64_t pixel) { return mix64(seedKey ^ (pixel * 0xD1B54A32D192ED03ull + 0x8CB92BA72F3D8DD7ull)); }
RTD_INLINE Rng stream_for(uint64_t pixelKey, uint32_t sample) {
    const uint64_t t = pixelKey + (2ull * sample + 1ull) * RTD_GOLDEN;
    const uint64_t a = mix64(t);
    const uint64_t b = mix64(t + RTD_GOLDEN); // = pixelKey + (2 * sample + 2) * GOLDEN, without a second 64-bit multiply
    Rng r;
    r.x = mod_m31((uint32_t) (a & 0xFFFFFFFFull)); // uint (rand.Next ()) < 2^31-1 (Float.fs:33-36)
    r.y = mod_m31((uint32_t) (a >> 32));
    r.z = mod_m31((uint32_t) (b & 0xFFFFFFFFull));
    r.w = mod_m31((uint32_t) (b >> 32));
    if ((r.x | r.y | r.z | r.w) == 0u) r.w = 1u;
    return r;
}

// UnitVector.random (Point.fs:49-59): cube sample -> normalise, retry when |v|^2 < 1e-8
RTD_INLINE V3 random_unit(Rng &r) {
    V3 out;
    for (;;) {
        double r1 = rng_get(r), r2 = rng_get(r), r3 = rng_get(r);
        V3 v = mk((2.0 * r1) - 1.0, (2.0 * r2) - 1.0, (2.0 * r3) - 1.0);
        if (unitise(v, out)) break;
    }
    return out;
}

// ---- Pixel.fs:136-151 ; colour = R | G<<8 | B<<16 in one register ------------------------------------------------
#define RTD_WHITE 0x00FFFFFFu
#define RTD_BLACK 0x00000000u
#define RTD_HOTPINK (205u | (105u << 8) | (180u << 16)) /* Pixel.fs:61-66 */
// x / 255 for x <= 255 * 255: (x * 0x8081) >> 23, exact on that range (checked for all 65 026 values) and the product stays below
// 2^32, so it is one full-rate 24-bit multiply and a shift instead of the quarter-rate v_mul_hi_u32 of the generic /255.
RTD_INLINE uint32_t div255(uint32_t x) { return __umul24(x, 0x8081u) >> 23; }
RTD_INLINE uint32_t pix_combine(uint32_t a, uint32_t b) { // Pixel.combine: int (a * b) / 255 per channel
    uint32_t r = div255(__umul24(a & 0xFFu, b & 0xFFu));
    uint32_t g = div255(__umul24((a >> 8) & 0xFFu, (b >> 8) & 0xFFu));
    uint32_t bl = div255(__umul24((a >> 16) & 0xFFu, (b >> 16) & 0xFFu));
    return r | (g << 8) | (bl << 16);
}
RTD_INLINE uint32_t round_byte(double v) { return ((uint32_t) (int32_t) rint(v)) & 0xFFu; } // Math.Round |> byte
RTD_INLINE uint32_t pix_darken(double albedo, uint32_t p) { // Pixel.darken
    uint32_t r = round_byte((double) (p & 0xFFu) * albedo);
    uint32_t g = round_byte((double) ((p >> 8) & 0xFFu) * albedo);
    uint32_t b = round_byte((double) ((p >> 16) & 0xFFu) * albedo);
    return r | (g << 8) | (b << 16);
}

// Math.Pow(x, 5.0) (Sphere.fs:290).  x^5 carried in double-double (error ~2^-100) and rounded once, i.e. the
// correctly rounded power.  The C runtime pow that .NET calls is within 1 ulp of it (glibc: equal on 99.92 % of
// [0,2], measured) -- tests/test_gpu_parity.py::test_pow5_is_the_correctly_rounded_power.
RTD_INLINE double pow5(double x) {
    double h2 = x * x;
    double l2 = fma(x, x, -h2);
    double h4 = h2 * h2;
    double l4 = fma(h2, h2, -h4) + 2.0 * (h2 * l2);
    double s4 = h4 + l4;
    double t4 = l4 - (s4 - h4);
    double h5 = s4 * x;
    double l5 = fma(s4, x, -h5) + t4 * x;
    return h5 + l5;
}

// ---- flattened scene ------------------------------------------------------------------------------------
// Image layout (bytes; every section 16-byte aligned; built by rt_scene.h; node, geo and meta are staged verbatim into LDS, mat is
// read from global memory -- only the general `reflection` of the rarer styles needs it):
//   node [n_nodes]  112 B  {lo,hi,hi,lo}_x {lo,hi,hi,lo}_y {lo,hi,hi,lo}_z : double; on_hit, on_miss, prim, pad : int32
//                          Each axis is stored {lo, hi, hi, lo} so that a ray's (near, far) pair is ONE aligned 16-byte read
//                          whichever way it travels: +0 for a non-negative inverse direction (lo, hi), +16 for a negative one
//                          (hi, lo) -- one address add and one ds_read_b128 per axis.  (ds_read_b128 moves 256 B per LDS clock,
//                          the ds_read2_b64 of the earlier {hi,lo,hi} layout 128: with 16 waves per CU in the node loop the LDS
//                          array was busy 64 % of the frame.)  A visit reads 6 of the 12 doubles and the links.
//                          on_miss = BYTE offset of the node to visit when this box is missed (n_nodes*112 = end);
//                          on_hit  = byte offset of the next record for a Branch, RTD_LEAF | the record's OWN offset for a
//                          Leaf: a walk that hits a leaf box stops there with the flag set, and the leaf test reads the
//                          sphere's index (prim) and where to go on (on_miss) from the record itself -- so a walk carries
//                          one position and nothing else.  prim = object index of a Leaf, -1 for a Branch.  The LDS copy
//                          holds ABSOLUTE LDS addresses in both links (patched when the image is staged), so a walk
//                          position is used as an address as it is.
//   geo  [n_obj][3]  d2    sphere {cx,cy}{cz,r^2}{albedo,radius} | plane {px,py}{pz,nx}{ny,nz}   48 B/object
//   meta [n_obj]     i2    {kind|style<<2|flipped<<5, rgb|(texture+1)<<24}                   8 B/object
//   node32 [n_nodes] 64 B  {lo,hi,hi,lo}_x {lo,hi,hi,lo}_y {lo,hi,hi,lo}_z : FLOAT; on_hit, on_miss, prim, pad : int32
//                          The same tree with every box rounded OUTWARD to single precision (lo down, hi up): the records of the
//                          timed kernel variant's node loop, which runs a conservative single-precision filter instead of the exact
//                          test (node_loop_lds32) and leaves the exact BoundingBox.hits to the leaf pass.  Links as in `node`, in
//                          units of 64 B.
//   mat  [n_obj][4]  double sphere {-, fuzz|ior, prob, 1/ior} (Dielectric) | {schlickInside, ior, 1/ior, schlickOutside} (Glass):
//                          per-material values of Sphere.fs:117,283-289; plane {albedo, fuzz, 0, 0}    32 B/object, global memory
// Objects: bounded spheres first (tree leaves point at them), then the unbounded list in Scene.make order.
// Order in the image: node | geo | meta | node32 | mat.  The counting kernel variant stages [node, meta end) into LDS, the timed
// one [geo, node32 end).
#ifndef RTD_NODE_BYTES
#define RTD_NODE_BYTES 112
#endif
#define RTD_NODE32_BYTES 64
#define RTD_LEAF 0x40000000 /* flag in on_hit / in a walk offset: a leaf's primitive test is pending */
struct TexRec { // global memory only
    uint32_t kind;
    uint32_t rgb;       // packed colour
    uint32_t ramp;      // src0 | src1<<8 | src2<<16
    int32_t even, odd;
    int32_t width, height;
    uint32_t texel_off; // byte offset into the texel blob
    double grid;
    double cx, cy, cz, map_radius; // interpret = Sphere.planeMapInverse map_radius (cx,cy,cz); 1.0/radius is formed on the device as Sphere.fs:57 does
};

template <bool LDS> struct Ptrs;
template <> struct Ptrs<true> {
    typedef const RTD_AS3 unsigned char *bp;
    typedef const RTD_AS3 d2 *d2p;
    typedef const RTD_AS3 i2 *i2p;
    typedef const RTD_AS3 int *ip;
    typedef const RTD_AS3 double *dp;
};
template <> struct Ptrs<false> {
    typedef const unsigned char *bp;
    typedef const d2 *d2p;
    typedef const i2 *i2p;
    typedef const int *ip;
    typedef const double *dp;
};

template <bool LDS> struct SceneView {
    typename Ptrs<LDS>::bp node;
    typename Ptrs<LDS>::d2p geo;
    typename Ptrs<LDS>::i2p meta;
    const double *mat; // global memory in both variants
    int n_nodes, n_bounded, n_unbounded;
    int first, end; // walk positions of the root record and of "tree exhausted" (LDS: absolute addresses; else offsets from `node`)
    int lds_lim;    // global-memory timed variant: node32 records at offsets below this are ALSO at the same offset in LDS (node_loop_glb32)
    int lds_thr;    // ... and a trip serves only the lanes at such records when there are at least this many of them (>= 1; 65: never)
    int narrow;     // fewer than 16384 objects: leaf queue entries are 16 bits, two to a word (always so for an LDS-resident scene)
    const TexRec *tex;
    const uint8_t *texels;
};

struct SceneOffsets { // byte offsets into the image
    uint32_t node, geo, meta, node32, mat, total;
    uint32_t lds_total;   // node + geo + meta: what the counting LDS variant of the kernel stages (from offset 0)
    uint32_t lds32_total; // geo + meta + node32: what the timed LDS variant stages (from offset `geo`)
    int32_t n_nodes, n_bounded, n_unbounded;
    float bmax;           // >= |every coordinate of every node32 box| and >= 1e-30: the scale of the filter's margin (walk_ctx32)
    int32_t box_implied;  // 1: every bounded sphere has 0 < radius <= 100, bmax <= 1000 and radius_max * bmax <= 500 (leaf_test_object_exact)
};

#define RTD_KIND_SPHERE 0u
#define RTD_KIND_PLANE 2u

struct Counters { uint32_t rays, aabb, prim, refl; };

// ---- BoundingBox.hits (BoundingBox.fs:30-94) ------------------------------------------------------------------
// Restated without branches, swaps or select chains; each step below is an identity on the reference's result:
//  * `if inv < 0 then swap t0 t1` (BoundingBox.fs:52-55): the caller passes near = (inv < 0 ? hi : lo) and
//    far = (inv < 0 ? lo : hi), so (near - o)*inv IS the swapped t0 and (far - o)*inv the swapped t1, same arithmetic.
//  * `tMin <- if t0 > tMin then t0 else tMin` == fmax(t0, tMin): tMin starts at -inf and is only ever replaced by a
//    t0 that compared greater, so it is never NaN; when t0 is NaN both forms keep tMin; on equal values they differ at
//    most in the sign of a zero, which no later comparison can see.  Likewise tMax and fmin.
//  * With tMin/tMax never NaN, `not (tMax < tMin || 0.0 >= tMax)` is `tMax >= tMin && tMax > 0.0`; tMin only rises and
//    tMax only falls over the three axes, so the x- and y-stage tests are implied by the later ones except `tMax_y > 0`
//    (the reference's asymmetry: bail-outs use `0.0 >= tMax`, the final test `tMax >= 0.0`).
//  * The final `tMax >= 0.0` is folded into tMin's starting value: with tMin' = max(tMin, 0), `tMax >= tMin'` is
//    `tMax >= tMin && tMax >= 0`.
RTD_INLINE bool bbox_hits_nf(double ix, double iy, double iz, V3 o, double nx, double fx, double ny, double fy, double nz, double fz) {
    double tMin = __builtin_fmax((nx - o.x) * ix, 0.0);            // NaN (0 * inf) -> 0.0, exactly as -inf would end up
    double tMax = __builtin_fmin((fx - o.x) * ix, __builtin_inf()); // NaN -> +inf
    tMin = __builtin_fmax((ny - o.y) * iy, tMin);
    tMax = __builtin_fmin((fy - o.y) * iy, tMax);
    const bool yPos = tMax > 0.0;
    tMin = __builtin_fmax((nz - o.z) * iz, tMin);
    tMax = __builtin_fmin((fz - o.z) * iz, tMax);
    return (tMax >= tMin) & yPos;
}
// (min,max)-ordered operands, as the unit hook and the reference's tests hand them over
RTD_INLINE bool bbox_hits(double ix, double iy, double iz, V3 o, d2 bx, d2 by, d2 bz) {
    return bbox_hits_nf(ix, iy, iz, o, ix < 0.0 ? bx.y : bx.x, ix < 0.0 ? bx.x : bx.y, iy < 0.0 ? by.y : by.x, iy < 0.0 ? by.x : by.y,
                        iz < 0.0 ? bz.y : bz.x, iz < 0.0 ? bz.x : bz.y);
}

// ---- Sphere.firstIntersection (Sphere.fs:349-386); returns NaN for ValueNone ----------------------------------------
// The Greater branch (Sphere.fs:362-380) restated without its Float.compare: i1 = fl(s - b) and i2 = -fl(b + s) with s > 1e-4.
// Rounding is monotonic and symmetric, so i1 >= i2 always; hence i2 positive implies i1 positive, and when both are positive
// `Float.compare i1 i2` is Greater (-> i2) unless |i1 - i2| < 1e-8, which -- the difference being ~2s > 2e-4 or, for |b| so
// large that s is absorbed, a multiple of ulp(b) > 1e-8 -- happens only when i1 == i2 exactly (-> i1, the same double).
// So the result is i2 if i2 > tol, else i1 if i1 > tol, else none; and the closing `Float.positive` of Sphere.fs:382-386 can
// only reject the Equal branch's -b.  The oracle keeps the reference's literal control flow; tests compare the two.
RTD_INLINE double sphere_first_intersection(V3 o, V3 d, V3 c, double r2) {
    V3 diff = vsub(o, c);
    double b = dot(d, diff);
    double cc = dot(diff, diff) - r2;
    double disc = (b * b - cc);
    int cmp = fcmp(disc, 0.0);
    if (cmp == CMP_EQ) { double i = (-b); return fpos(i) ? i : __builtin_nan(""); }
    double i = __builtin_nan("");
    if (cmp == CMP_GT) { // also taken by a NaN discriminant, which yields NaN roots and so none
        double s = sqrt_above_tol(disc); // disc > 1e-8 in this branch
        double i1 = s - b;
        double i2 = -(b + s);
        i = fpos(i2) ? i2 : (fpos(i1) ? i1 : i);
    }
    return i;
}

// ---- InfinitePlane.intersection (InfinitePlane.fs:125-136) -----------------------------------------------------
RTD_INLINE double plane_intersection(V3 o, V3 d, V3 p0, V3 n) {
    double den = dot(n, d);
    if (feq(den, 0.0)) return __builtin_nan("");
    double t = dot(n, vsub(p0, o)) / den;
    return fpos(t) ? t : __builtin_nan("");
}

// ---- Scene.hitObject (Scene.fs:62-91) + Scene.bestCandidate (Scene.fs:30-60) ------------------------------------------
// The recursive left-then-right walk becomes a stackless loop over the pre-order image: a hit box advances to the next
// record (its left child, or -- for a leaf -- whatever follows it), a missed box jumps to its skip offset.
// The walk is split into resumable pieces so that the render kernel can interleave the walks of its 64 lanes with the
// other stages of their paths (rt_render_kernel.h): `Walk` is the state a lane keeps while parked, `WalkCtx` is derived
// from the ray each time a lane (re)enters the node loop.  The order in which a ray sees its leaves never changes, so the
// strict-`<` tie-breaking of Scene.fs:45-47 is unchanged.
struct Walk {
    int off;        // position of the next node record (SceneView::first ..); >= SceneView::end when the tree is exhausted or the lane is not walking;
                    // RTD_LEAF|position of a Leaf record while that leaf's primitive test is pending
    int best;       // bestObject (object index) or -1
    double bestLen; // bestLength; NaN until something is hit (Scene.fs:64)
};
struct WalkCtx {
    double ix, iy, iz; // BoundingBox.inverseDirections (BoundingBox.fs:25-28)
    int nX, nY, nZ;    // 0 or 16: byte offset of each axis' (near, far) pair inside its {lo, hi, hi, lo} quad: the swap of BoundingBox.fs:52-55
    double bestF;      // bestFloat = bestLength^2, +inf until something is hit (Scene.fs:65)
};
RTD_INLINE void walk_begin(Walk &w, int first) { w.off = first; w.best = -1; w.bestLen = __builtin_nan(""); }
RTD_INLINE WalkCtx walk_ctx(V3 d, const Walk &w) {
    WalkCtx c;
    c.ix = 1.0 / d.x; c.iy = 1.0 / d.y; c.iz = 1.0 / d.z;
    c.nX = c.ix < 0.0 ? 16 : 0; c.nY = c.iy < 0.0 ? 16 : 0; c.nZ = c.iz < 0.0 ? 16 : 0;
    c.bestF = (w.best < 0) ? __builtin_inf() : w.bestLen * w.bestLen; // `a = point * point` (Scene.fs:45), recomputed
    return c;
}
template <bool LDS> RTD_INLINE typename Ptrs<LDS>::bp node_at(const SceneView<LDS> &sc, int pos);
template <> RTD_INLINE Ptrs<true>::bp node_at<true>(const SceneView<true> &, int pos) { return (Ptrs<true>::bp) (uintptr_t) (uint32_t) pos; }
template <> RTD_INLINE Ptrs<false>::bp node_at<false>(const SceneView<false> &sc, int pos) { return sc.node + pos; }
// One BoundingBox.hits + advance: w.off becomes on_hit or on_miss; a hit Leaf leaves RTD_LEAF|its own position there.
template <bool LDS>
RTD_INLINE void node_step(const SceneView<LDS> &sc, V3 o, const WalkCtx &c, Walk &w) {
    typedef typename Ptrs<LDS>::bp bp;
    typedef typename Ptrs<LDS>::i2p i2p;
    bp rec = node_at<LDS>(sc, w.off);
    typedef typename Ptrs<LDS>::d2p d2p;
    const d2 vx = *(d2p) (rec + c.nX), vy = *(d2p) (rec + 32 + c.nY), vz = *(d2p) (rec + 64 + c.nZ);
    const double vnx = vx.x, vfx = vx.y, vny = vy.x, vfy = vy.y, vnz = vz.x, vfz = vz.y;
    const i2 lk = *(i2p) (rec + 96);
    const bool hit = bbox_hits_nf(c.ix, c.iy, c.iz, o, vnx, vfx, vny, vfy, vnz, vfz);
    w.off = hit ? lk.x : lk.y;
}
// The node loop of the LDS-resident scene, as one block of assembly: node_step for every lane with off < end, again and again
// until at most `stop` lanes are still walking.  The instructions are those the compiler makes of node_step + bbox_hits_nf
// (same operations, same operand order, hence the same bits); what the compiler could not be talked into is ONE loop-carried
// register (the walk position) with no copies of it -- the compiled loop shuffled it through five to eight v_mov per trip
// (1.8e9 trips per bench frame).  v[100:115] hold the three (near, far) pairs and the four link words (the halves of a 128-bit
// operand cannot be named through an asm operand, so these are fixed registers, declared as clobbers).
// Lanes at >= end (finished, idle) or with a full queue are masked off; exec is restored before the block ends.
// A hit Leaf does not stop its lane: the walk goes on (in the LDS copy a Leaf's on_hit link is its on_miss link, stage_scene) and the
// leaf's object goes into the lane's QUEUE of pending sphere tests -- `pend`, two 16-bit entries `0x4000 | object`, the newer in the
// high half -- with one v_alignbit: the record's third link word holds the entry, its fourth the shift (16 for a Leaf, 0 for a
// Branch; a miss shifts by 0 whatever the record says).  A lane whose queue is full (low half occupied) sits out until the leaf pass
// outside has taken its older entry.  Measured with the queue built into the compiled loop of the counting variant: node trips
// 1.355e9 -> 1.241e9 per frame, leaf passes 1.995e8 -> 1.21e8 (the lanes stop half as often, and the passes find more lanes to serve).
RTD_INLINE int node_loop_lds(int off, uint32_t &pend, int end, int stop, V3 o, const WalkCtx &c) {
    int ax, ay, az, cnt;
    unsigned long long save, save2;
    const double inf = __builtin_inf();
    asm volatile(
        "s_waitcnt lgkmcnt(0)\n"
        "1:\n"
        "  v_cmp_gt_i32 vcc, %[end], %[off]\n"
        "  v_cmp_eq_u16 %[save2], 0, %[pend]\n"
        "  s_and_b64 vcc, vcc, %[save2]\n"
        "  s_bcnt1_i32_b64 %[cnt], vcc\n"
        "  s_cmp_le_u32 %[cnt], %[stop]\n"
        "  s_cbranch_scc1 2f\n"
        "  s_and_saveexec_b64 %[save], vcc\n"
        "  v_add_u32 %[ax], %[off], %[nx]\n"
        "  v_add_u32 %[ay], %[off], %[ny]\n"
        "  v_add_u32 %[az], %[off], %[nz]\n"
        "  ds_read_b128 v[100:103], %[ax]\n"
        "  ds_read_b128 v[104:107], %[ay] offset:32\n"
        "  ds_read_b128 v[108:111], %[az] offset:64\n"
        "  ds_read_b128 v[112:115], %[off] offset:96\n"
        "  s_waitcnt lgkmcnt(3)\n"
        "  v_add_f64 v[100:101], v[100:101], -%[ox]\n"
        "  v_mul_f64 v[100:101], %[ix], v[100:101]\n"
        "  v_add_f64 v[102:103], v[102:103], -%[ox]\n"
        "  v_max_f64 v[100:101], v[100:101], 0\n"
        "  v_mul_f64 v[102:103], %[ix], v[102:103]\n"
        "  v_min_f64 v[102:103], v[102:103], %[inf]\n"
        "  s_waitcnt lgkmcnt(2)\n"
        "  v_add_f64 v[104:105], v[104:105], -%[oy]\n"
        "  v_mul_f64 v[104:105], %[iy], v[104:105]\n"
        "  v_max_f64 v[100:101], v[104:105], v[100:101]\n"
        "  v_add_f64 v[106:107], v[106:107], -%[oy]\n"
        "  v_mul_f64 v[106:107], %[iy], v[106:107]\n"
        "  v_min_f64 v[102:103], v[106:107], v[102:103]\n"
        "  s_waitcnt lgkmcnt(1)\n"
        "  v_add_f64 v[108:109], v[108:109], -%[oz]\n"
        "  v_mul_f64 v[108:109], %[iz], v[108:109]\n"
        "  v_max_f64 v[100:101], v[108:109], v[100:101]\n"
        "  v_add_f64 v[110:111], v[110:111], -%[oz]\n"
        "  v_mul_f64 v[110:111], %[iz], v[110:111]\n"
        "  v_cmp_lt_f64 vcc, 0, v[102:103]\n"
        "  v_min_f64 v[102:103], v[110:111], v[102:103]\n"
        "  v_cmp_ge_f64 %[save2], v[102:103], v[100:101]\n"
        "  s_and_b64 vcc, vcc, %[save2]\n"
        "  s_waitcnt lgkmcnt(0)\n"
        "  v_cndmask_b32 %[off], v113, v112, vcc\n"
        "  v_cndmask_b32 v113, 0, v115, vcc\n"
        "  v_alignbit_b32 %[pend], v114, %[pend], v113\n"
        "  s_mov_b64 exec, %[save]\n"
        "  s_branch 1b\n"
        "2:\n"
        : [off] "+v"(off), [pend] "+v"(pend), [ax] "=&v"(ax), [ay] "=&v"(ay), [az] "=&v"(az), [cnt] "=&s"(cnt), [save] "=&s"(save), [save2] "=&s"(save2)
        : [end] "s"(end), [stop] "s"(stop), [nx] "v"(c.nX), [ny] "v"(c.nY), [nz] "v"(c.nZ), [ox] "v"(o.x), [oy] "v"(o.y), [oz] "v"(o.z),
          [ix] "v"(c.ix), [iy] "v"(c.iy), [iz] "v"(c.iz), [inf] "s"(inf)
        : "vcc", "scc", "memory", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113",
          "v114", "v115");
    return off;
}
#define RTD_PEND_MARK 0x4000u /* a queue entry is RTD_PEND_MARK | object: never zero, objects of an LDS-resident scene are < 16384 */

// ---- the node loop of the timed variant: a conservative single-precision FILTER, the exact test left to the leaf pass -----------
// Scene.bestCandidate tests a sphere iff the ray hits its Leaf box and every Branch box above it, and a ray that hits a Leaf box
// hits every box that contains it (rt_scene.h, "the tree the device WALKS"): the spheres tested for a ray are {i : hits leafBox_i},
// whatever stands above the leaves.  So the walk above the leaves may use ANY test F with  hits(box) => F(box)  for every box:
//     tested set = {i : F(every box above leaf i) && F(leafBox_i) && hits(leafBox_i)} = {i : hits(leafBox_i)},
// because hits(leafBox_i) => hits(every box above it) => F(every box above it).  F's false positives only cost visits.
// F here: the slab test in single precision on boxes rounded OUTWARD (lo down, hi up; rt_scene.h) with every per-axis entry
// distance pushed down and every exit distance pushed up by a margin that covers all rounding between the two computations:
//     exact:   t = fl64(fl64(b - o) * fl64(1/d))                           = (b - o)/d * (1 + th),  |th| <= 3.1 * 2^-53
//     filter:  t' = fl32(b32 * i + c),  i = rcp32(fl32(d)) = (1/d)(1 + k), |k| <= 3.01 e  (e = 2^-24; v_rcp_f32 is within 1 ulp)
//              oi = fl32(fl32(o) * i) = (o/d)(1 + l), |l| <= 5.02 e;   c = fl32(-oi -+ m)
//     => |t' - (b32 - o)/d| <= 4.1 e |b32/d| + 7.1 e |o/d| + underflow (< 1e-37)   when m = 0,
// and (b32 - o)/d is on the safe side of (b - o)/d by the outward rounding (near: lo for d >= 0, hi for d < 0 -- the exact test's
// own choice, BoundingBox.fs:52-55, keyed on the sign bit of d like the sign of fl64(1/d)).  The margin is
//     m = fl32(fma(|i|, bmax, |oi|) * 2^-20 + 1e-30)  >=  16 e (|b32/d| + |o/d|)(1 - 8 e) + 1e-30        (bmax >= |b32| for every box),
// more than twice what is needed.  Infinities and NaNs: +-inf or NaN anywhere in the chain makes m = inf or NaN, hence a lower
// bound of -inf or NaN and an upper bound of +inf or NaN; v_max3/v_min3 return the non-NaN operand, v_max(x, 0) is never NaN, and
// the final test is `not (tmax < tmin)`, true for a NaN tmax: such an axis (or ray) constrains nothing, as in the exact test,
// whose NaN products leave tMin/tMax as they were.  The exact BoundingBox.hits of a Leaf box runs in the leaf pass
// (leaf_test_object_exact), so no sphere is tested that the reference would not test.
// tests: test_filter_is_conservative (CPU model, 1e8 cases incl. rays through edges and corners, origins on faces, axis-aligned and
// denormal directions, huge and tiny boxes) and, on the GPU, rt_dev_bbox_filter over the same generators plus every render test
// (the counting variant walks the exact double-precision records; every render test compares the two variants).
struct WalkCtx32 {
    float ix, iy, iz;    // ~ 1 / d
    float cnx, cny, cnz; // entry distance of axis a is bounded below by fma(near32_a, i_a, cn_a)
    float cfx, cfy, cfz; // exit distance above by fma(far32_a, i_a, cf_a)
    int nX, nY, nZ;      // 0 or 8: byte offset of the (near, far) pair inside the axis' {lo, hi, hi, lo} quad of floats
};
RTD_INLINE void filter_axis(double o, double d, float bmax, float &inv, float &cn, float &cf, int &nOff) {
    inv = __builtin_amdgcn_rcpf((float) d);
    const float oi = (float) o * inv;
    const float m = fmaf(__builtin_fabsf(inv), bmax, __builtin_fabsf(oi)) * 0x1p-20f + 1e-30f;
    cn = -oi - m;
    cf = m - oi;
    nOff = (int) (((uint32_t) __double2hiint(d) >> 28) & 8u); // the sign bit of d, as 0 or 8
}
RTD_INLINE WalkCtx32 walk_ctx32(V3 o, V3 d, float bmax) {
    WalkCtx32 c;
    filter_axis(o.x, d.x, bmax, c.ix, c.cnx, c.cfx, c.nX);
    filter_axis(o.y, d.y, bmax, c.iy, c.cny, c.cfy, c.nY);
    filter_axis(o.z, d.z, bmax, c.iz, c.cnz, c.cfz, c.nZ);
    return c;
}
// F for one box given as single-precision (lo, hi) pairs already rounded outward (unit hook and the CPU model's counterpart)
RTD_INLINE bool bbox_filter(const WalkCtx32 &c, float lox, float hix, float loy, float hiy, float loz, float hiz) {
    const float tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(fmaf(c.nX ? hix : lox, c.ix, c.cnx), fmaf(c.nY ? hiy : loy, c.iy, c.cny)),
                                                     fmaf(c.nZ ? hiz : loz, c.iz, c.cnz)), 0.0f);
    const float tf = __builtin_fminf(__builtin_fminf(fmaf(c.nX ? lox : hix, c.ix, c.cfx), fmaf(c.nY ? loy : hiy, c.iy, c.cfy)), fmaf(c.nZ ? loz : hiz, c.iz, c.cfz));
    return !(tf < tn);
}
// The loop itself, same shape as node_loop_lds: 18 VALU + 4 LDS instructions per visit, none of them double precision
// (3 address adds, 3 ds_read_b64 + 1 ds_read_b128, 6 v_fma_f32, max3 / min3 / max, 1 compare, 2 selects, 1 v_alignbit, 2 for the
// activity test) -- measured issue cost ~80 cycles per visit against ~147 for the double-precision loop (profiles/r3/valu_rates.txt).
// Diagnostic builds (-DRTD_STAGE_CLOCKS) count the loop's trips and the lanes that step in them (StageStats::loopTrips, loopLanes)
#ifdef RTD_STAGE_CLOCKS
#define RTD_TRIP_COUNT(S) S
#define RTD_TRIP_ARGS , unsigned &trips, unsigned &lanes
#define RTD_TRIP_OPS , [trips] "+s"(trips), [lanes] "+s"(lanes)
#else
#define RTD_TRIP_COUNT(S)
#define RTD_TRIP_ARGS
#define RTD_TRIP_OPS
#endif
RTD_INLINE int node_loop_lds32(int off, uint32_t &pend, int end, int stop, const WalkCtx32 &c RTD_TRIP_ARGS) {
    int ax, ay, az, cnt;
    unsigned long long save, save2;
#ifdef RTD_PK_FMA
    auto pk = [](float lo, float hi) { return ((unsigned long long) __float_as_uint(hi) << 32) | (unsigned long long) __float_as_uint(lo); };
    const unsigned long long pix = pk(c.ix, c.ix), piy = pk(c.iy, c.iy), piz = pk(c.iz, c.iz);
    const unsigned long long pcx = pk(c.cnx, c.cfx), pcy = pk(c.cny, c.cfy), pcz = pk(c.cnz, c.cfz);
#endif
    asm volatile(
        "s_waitcnt lgkmcnt(0)\n"
        "1:\n"
        "  v_cmp_gt_i32 vcc, %[end], %[off]\n"
        "  v_cmp_eq_u16 %[save2], 0, %[pend]\n"
        "  s_and_b64 vcc, vcc, %[save2]\n"
        "  s_bcnt1_i32_b64 %[cnt], vcc\n"
        "  s_cmp_le_u32 %[cnt], %[stop]\n"
        "  s_cbranch_scc1 2f\n"
        "  s_and_saveexec_b64 %[save], vcc\n"
        RTD_TRIP_COUNT("  s_add_u32 %[trips], %[trips], 1\n  s_add_u32 %[lanes], %[lanes], %[cnt]\n")
        "  v_add_u32 %[ax], %[off], %[nx]\n"
        "  v_add_u32 %[ay], %[off], %[ny]\n"
        "  v_add_u32 %[az], %[off], %[nz]\n"
        "  ds_read_b64 v[100:101], %[ax]\n"
        "  ds_read_b64 v[102:103], %[ay] offset:16\n"
        "  ds_read_b64 v[104:105], %[az] offset:32\n"
        "  ds_read_b128 v[106:109], %[off] offset:48\n"
        "  s_waitcnt lgkmcnt(3)\n"
#ifdef RTD_PK_FMA /* (near, far) of an axis in one packed instruction: the inverse direction's low half serves both */
        "  v_pk_fma_f32 v[100:101], v[100:101], %[pix], %[pcx] op_sel_hi:[1,0,1]\n"
        "  s_waitcnt lgkmcnt(2)\n"
        "  v_pk_fma_f32 v[102:103], v[102:103], %[piy], %[pcy] op_sel_hi:[1,0,1]\n"
        "  s_waitcnt lgkmcnt(1)\n"
        "  v_pk_fma_f32 v[104:105], v[104:105], %[piz], %[pcz] op_sel_hi:[1,0,1]\n"
#else
        "  v_fma_f32 v100, v100, %[ix], %[cnx]\n"
        "  v_fma_f32 v101, v101, %[ix], %[cfx]\n"
        "  s_waitcnt lgkmcnt(2)\n"
        "  v_fma_f32 v102, v102, %[iy], %[cny]\n"
        "  v_fma_f32 v103, v103, %[iy], %[cfy]\n"
        "  s_waitcnt lgkmcnt(1)\n"
        "  v_fma_f32 v104, v104, %[iz], %[cnz]\n"
        "  v_fma_f32 v105, v105, %[iz], %[cfz]\n"
#endif
        "  v_max3_f32 v100, v100, v102, v104\n"
        "  v_min3_f32 v101, v101, v103, v105\n"
        "  v_max_f32 v100, 0, v100\n"
        "  v_cmp_nlt_f32 vcc, v101, v100\n"
        "  s_waitcnt lgkmcnt(0)\n"
        "  v_cndmask_b32 %[off], v107, v106, vcc\n"
        "  v_cndmask_b32 v107, 0, v109, vcc\n"
        "  v_alignbit_b32 %[pend], v108, %[pend], v107\n"
        "  s_mov_b64 exec, %[save]\n"
        "  s_branch 1b\n"
        "2:\n"
        : [off] "+v"(off), [pend] "+v"(pend), [ax] "=&v"(ax), [ay] "=&v"(ay), [az] "=&v"(az), [cnt] "=&s"(cnt), [save] "=&s"(save), [save2] "=&s"(save2) RTD_TRIP_OPS
        : [end] "s"(end), [stop] "s"(stop), [nx] "v"(c.nX), [ny] "v"(c.nY), [nz] "v"(c.nZ), [ix] "v"(c.ix), [iy] "v"(c.iy), [iz] "v"(c.iz),
          [cnx] "v"(c.cnx), [cny] "v"(c.cny), [cnz] "v"(c.cnz), [cfx] "v"(c.cfx), [cfy] "v"(c.cfy), [cfz] "v"(c.cfz)
#ifdef RTD_PK_FMA
          , [pix] "v"(pix), [piy] "v"(piy), [piz] "v"(piz), [pcx] "v"(pcx), [pcy] "v"(pcy), [pcz] "v"(pcz)
#endif
        : "vcc", "scc", "memory", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109");
    return off;
}
// The oldest entry of a lane's queue (the low half if occupied, else the high half), removed from it.
RTD_INLINE int pend_pop(uint32_t &pend) {
    const uint32_t lo = pend & 0xFFFFu;
    const uint32_t e = lo != 0u ? lo : (pend >> 16);
    pend = lo != 0u ? (pend & 0xFFFF0000u) : 0u;
    return (int) (e & (RTD_PEND_MARK - 1u));
}
// Leaf: Hittable.hits (Hittable.fs:27-31) -> Sphere.firstIntersection, kept if t^2 < bestFloat (strict; NaN fails).
// On an exact tie the reference keeps the leaf its depth-first walk met first; object indices are those ranks (rt_scene.h),
// so the second clause reproduces that when the walked tree visits leaves in another order (and never fires when it does not).
template <bool LDS>
RTD_INLINE void leaf_test(const SceneView<LDS> &sc, V3 o, V3 d, WalkCtx &c, Walk &w) {
    typedef typename Ptrs<LDS>::bp bp;
    bp rec = node_at<LDS>(sc, w.off & (RTD_LEAF - 1));
    const int next = *(typename Ptrs<LDS>::ip) (rec + 100), prim = *(typename Ptrs<LDS>::ip) (rec + 104); // on_miss, prim
    const d2 g0 = sc.geo[prim * 3 + 0], g1 = sc.geo[prim * 3 + 1];
    const double t = sphere_first_intersection(o, d, mk(g0.x, g0.y, g1.x), g1.y);
    const double a = t * t;
    if (a < c.bestF || (a == c.bestF && prim < w.best)) { c.bestF = a; w.best = prim; w.bestLen = t; }
    w.off = next;
}
// The same for a queued leaf: its object index is all that is needed (the walk has long moved on)
template <bool LDS>
RTD_INLINE void leaf_test_object(const SceneView<LDS> &sc, V3 o, V3 d, WalkCtx &c, Walk &w, int prim) {
    const d2 g0 = sc.geo[prim * 3 + 0], g1 = sc.geo[prim * 3 + 1];
    const double t = sphere_first_intersection(o, d, mk(g0.x, g0.y, g1.x), g1.y);
    const double a = t * t;
    if (a < c.bestF || (a == c.bestF && prim < w.best)) { c.bestF = a; w.best = prim; w.bestLen = t; }
}
// The same filter loop for scenes that do not fit the LDS: `off` is the record's byte offset from the start of the node32 section
// (the links as the host stored them).  The section is ordered by depth (rt_scene.h), and its first `lim` bytes -- as many records
// as the LDS has room for -- were copied to the START of the workgroup's LDS (stage_nodes32): a visit reads its record from LDS if
// off < lim, from global memory (scalar base + offset) otherwise.  A trip that reads global memory waits ~3 times as long as one that
// reads LDS, and waits as a whole: so while `thr` or more lanes stand at records in LDS the trip is theirs alone and the lanes at
// deeper records wait where they are; with fewer, one trip serves both kinds (reads issued under complementary masks, awaited
// together).  Every trip advances at least one lane (thr >= 1).  Measured on the bench scene's recipe with 1026 / 2705 / 6399 spheres
// (tuned trees): 18.5 / 20.4 / 20.9 ms with every record in global memory, 17.3 / 18.9 / 19.3 with the split but every trip serving
// all lanes, 14.3 / 15.7 / 16.2 with thr = 16 (8..24 within 2 %; a second threshold on the number of waiting lanes gave nothing).  The queue of
// pending leaves is two FULL-WIDTH entries (pend0 the older, pend1 the newer; an entry is the record's third link word,
// RTD_PEND_MARK | object for scenes of < 16384 objects, RTD_PEND_WIDE | object beyond): such scenes may have millions of objects.
#define RTD_PEND_WIDE 0x80000000u
RTD_INLINE int node_loop_glb32(int off, uint32_t &pend0, uint32_t &pend1, const unsigned char *base, int lim, int thr, int end, int stop, const WalkCtx32 &c) {
    int ax, ay, az, cnt;
    unsigned long long save, save2, save3;
    asm volatile(
        "s_waitcnt vmcnt(0) lgkmcnt(0)\n"
        "1:\n"
        "  v_cmp_gt_i32 vcc, %[end], %[off]\n"
        "  v_cmp_eq_u32 %[save2], 0, %[pend1]\n"
        "  s_and_b64 vcc, vcc, %[save2]\n"
        "  s_bcnt1_i32_b64 %[cnt], vcc\n"
        "  s_cmp_le_u32 %[cnt], %[stop]\n"
        "  s_cbranch_scc1 2f\n"
        "  s_and_saveexec_b64 %[save], vcc\n"              /* exec = the lanes that step */
        "  v_add_u32 %[ax], %[off], %[nx]\n"
        "  v_add_u32 %[ay], %[off], %[ny]\n"
        "  v_add_u32 %[az], %[off], %[nz]\n"
        "  v_cmp_gt_i32 vcc, %[lim], %[off]\n"             /* of those, the ones whose record is in LDS (vcc is 0 for the others) */
        "  s_bcnt1_i32_b64 %[cnt], vcc\n"
        "  s_cmp_ge_u32 %[cnt], %[thr]\n"
        "  s_cbranch_scc1 3f\n"
        "  s_and_saveexec_b64 %[save2], vcc\n"             /* save2 = the stepping lanes; exec = stepping & in LDS */
        "  ds_read_b64 v[100:101], %[ax]\n"
        "  ds_read_b64 v[102:103], %[ay] offset:16\n"
        "  ds_read_b64 v[104:105], %[az] offset:32\n"
        "  ds_read_b128 v[106:109], %[off] offset:48\n"
        "  s_andn2_b64 exec, %[save2], exec\n"             /* stepping & not in LDS */
        "  global_load_dwordx2 v[100:101], %[ax], %[base]\n"
        "  global_load_dwordx2 v[102:103], %[ay], %[base] offset:16\n"
        "  global_load_dwordx2 v[104:105], %[az], %[base] offset:32\n"
        "  global_load_dwordx4 v[106:109], %[off], %[base] offset:48\n"
        "  s_mov_b64 exec, %[save2]\n"
        "  s_branch 4f\n"
        "3:\n"                                             /* thr or more lanes can step out of LDS: a trip for those alone, the */
        "  s_mov_b64 exec, vcc\n"                          /* others wait where they are (no trip waits for global memory then) */
        "  ds_read_b64 v[100:101], %[ax]\n"
        "  ds_read_b64 v[102:103], %[ay] offset:16\n"
        "  ds_read_b64 v[104:105], %[az] offset:32\n"
        "  ds_read_b128 v[106:109], %[off] offset:48\n"
        "4:\n"
        "  s_waitcnt vmcnt(0) lgkmcnt(0)\n"
        "  v_fma_f32 v100, v100, %[ix], %[cnx]\n"
        "  v_fma_f32 v101, v101, %[ix], %[cfx]\n"
        "  v_fma_f32 v102, v102, %[iy], %[cny]\n"
        "  v_fma_f32 v103, v103, %[iy], %[cfy]\n"
        "  v_fma_f32 v104, v104, %[iz], %[cnz]\n"
        "  v_fma_f32 v105, v105, %[iz], %[cfz]\n"
        "  v_max3_f32 v100, v100, v102, v104\n"
        "  v_min3_f32 v101, v101, v103, v105\n"
        "  v_max_f32 v100, 0, v100\n"
        "  v_cmp_nlt_f32 vcc, v101, v100\n"
        "  v_cndmask_b32 %[off], v107, v106, vcc\n"
        "  v_cndmask_b32 v108, 0, v108, vcc\n"           /* the entry of a hit Leaf, else 0 */
        "  v_cmp_ne_u32 %[save3], 0, v108\n"
        "  v_cmp_eq_u32 vcc, 0, %[pend0]\n"
        "  s_and_b64 vcc, %[save3], vcc\n"               /* into the empty older slot */
        "  v_cndmask_b32 %[pend0], %[pend0], v108, vcc\n"
        "  s_andn2_b64 vcc, %[save3], vcc\n"             /* or into the newer one */
        "  v_cndmask_b32 %[pend1], %[pend1], v108, vcc\n"
        "  s_mov_b64 exec, %[save]\n"
        "  s_branch 1b\n"
        "2:\n"
        : [off] "+v"(off), [pend0] "+v"(pend0), [pend1] "+v"(pend1), [ax] "=&v"(ax), [ay] "=&v"(ay), [az] "=&v"(az), [cnt] "=&s"(cnt), [save] "=&s"(save),
          [save2] "=&s"(save2), [save3] "=&s"(save3)
        : [end] "s"(end), [lim] "s"(lim), [thr] "s"(thr), [stop] "s"(stop), [base] "s"(base), [nx] "v"(c.nX), [ny] "v"(c.nY), [nz] "v"(c.nZ), [ix] "v"(c.ix), [iy] "v"(c.iy), [iz] "v"(c.iz),
          [cnx] "v"(c.cnx), [cny] "v"(c.cny), [cnz] "v"(c.cnz), [cfx] "v"(c.cfx), [cfy] "v"(c.cfy), [cfz] "v"(c.cfz)
        : "vcc", "scc", "memory", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109");
    return off;
}
// The same loop for scenes of fewer than 16384 objects, whose queue entries fit 16 bits: the queue is node_loop_lds32's (two entries in
// one word, pushed by v_alignbit_b32; `pend1` is not the loop's), which takes six vector and as many scalar instructions out of a
// visit; and a trip that reads only LDS overlaps its arithmetic with the reads as node_loop_lds32 does.  Measured with every record
// of a 1026-sphere scene's tree in LDS: 191 SIMD cycles per trip through node_loop_glb32 against 90 through node_loop_lds32 -- the
// instructions around the box test, not the reads, were what such scenes paid for.
RTD_INLINE int node_loop_hyb16(int off, uint32_t &pend, const unsigned char *base, int lim, int thr, int end, int stop, const WalkCtx32 &c RTD_TRIP_ARGS) {
    int ax, ay, az, cnt;
    unsigned long long save, save2;
#define RTD_HYB_TAIL                                                                                                                  \
        "  v_max3_f32 v100, v100, v102, v104\n"                                                                                       \
        "  v_min3_f32 v101, v101, v103, v105\n"                                                                                       \
        "  v_max_f32 v100, 0, v100\n"                                                                                                 \
        "  v_cmp_nlt_f32 vcc, v101, v100\n"                                                                                           \
        "  s_waitcnt vmcnt(0) lgkmcnt(0)\n"                                                                                           \
        "  v_cndmask_b32 %[off], v107, v106, vcc\n"                                                                                   \
        "  v_cndmask_b32 v107, 0, v109, vcc\n"                                                                                        \
        "  v_alignbit_b32 %[pend], v108, %[pend], v107\n"                                                                             \
        "  s_mov_b64 exec, %[save]\n"                                                                                                 \
        "  s_branch 1b\n"
    asm volatile(
        "s_waitcnt vmcnt(0) lgkmcnt(0)\n"
        "1:\n"
        "  v_cmp_gt_i32 vcc, %[end], %[off]\n"
        "  v_cmp_eq_u16 %[save2], 0, %[pend]\n"
        "  s_and_b64 vcc, vcc, %[save2]\n"
        "  s_bcnt1_i32_b64 %[cnt], vcc\n"
        "  s_cmp_le_u32 %[cnt], %[stop]\n"
        "  s_cbranch_scc1 2f\n"
        "  s_and_saveexec_b64 %[save], vcc\n"              /* exec = the lanes that step */
        RTD_TRIP_COUNT("  s_add_u32 %[trips], %[trips], 1\n")
        "  v_cmp_gt_i32 vcc, %[lim], %[off]\n"             /* of those, the ones whose record is in LDS (vcc is 0 for the others) */
        "  v_add_u32 %[ax], %[off], %[nx]\n"
        "  v_add_u32 %[ay], %[off], %[ny]\n"
        "  v_add_u32 %[az], %[off], %[nz]\n"
        "  s_bcnt1_i32_b64 %[cnt], vcc\n"
        "  s_cmp_ge_u32 %[cnt], %[thr]\n"
        "  s_cbranch_scc1 3f\n"
        "  s_and_saveexec_b64 %[save2], vcc\n"             /* save2 = the stepping lanes; exec = stepping & in LDS */
        "  ds_read_b64 v[100:101], %[ax]\n"
        "  ds_read_b64 v[102:103], %[ay] offset:16\n"
        "  ds_read_b64 v[104:105], %[az] offset:32\n"
        "  ds_read_b128 v[106:109], %[off] offset:48\n"
        "  s_andn2_b64 exec, %[save2], exec\n"             /* stepping & not in LDS */
        "  global_load_dwordx2 v[100:101], %[ax], %[base]\n"
        "  global_load_dwordx2 v[102:103], %[ay], %[base] offset:16\n"
        "  global_load_dwordx2 v[104:105], %[az], %[base] offset:32\n"
        "  global_load_dwordx4 v[106:109], %[off], %[base] offset:48\n"
        "  s_mov_b64 exec, %[save2]\n"
        RTD_TRIP_COUNT("  s_bcnt1_i32_b64 %[cnt], %[save2]\n  s_add_u32 %[lanes], %[lanes], %[cnt]\n")
        "  s_waitcnt vmcnt(0) lgkmcnt(0)\n"
        "  v_fma_f32 v100, v100, %[ix], %[cnx]\n"
        "  v_fma_f32 v101, v101, %[ix], %[cfx]\n"
        "  v_fma_f32 v102, v102, %[iy], %[cny]\n"
        "  v_fma_f32 v103, v103, %[iy], %[cfy]\n"
        "  v_fma_f32 v104, v104, %[iz], %[cnz]\n"
        "  v_fma_f32 v105, v105, %[iz], %[cfz]\n"
        RTD_HYB_TAIL
        "3:\n"                                             /* thr or more lanes can step out of LDS: a trip for those alone, the */
        RTD_TRIP_COUNT("  s_add_u32 %[lanes], %[lanes], %[cnt]\n")
        "  s_mov_b64 exec, vcc\n"                          /* others wait where they are (no trip waits for global memory then) */
        "  ds_read_b64 v[100:101], %[ax]\n"
        "  ds_read_b64 v[102:103], %[ay] offset:16\n"
        "  ds_read_b64 v[104:105], %[az] offset:32\n"
        "  ds_read_b128 v[106:109], %[off] offset:48\n"
        "  s_waitcnt lgkmcnt(3)\n"
        "  v_fma_f32 v100, v100, %[ix], %[cnx]\n"
        "  v_fma_f32 v101, v101, %[ix], %[cfx]\n"
        "  s_waitcnt lgkmcnt(2)\n"
        "  v_fma_f32 v102, v102, %[iy], %[cny]\n"
        "  v_fma_f32 v103, v103, %[iy], %[cfy]\n"
        "  s_waitcnt lgkmcnt(1)\n"
        "  v_fma_f32 v104, v104, %[iz], %[cnz]\n"
        "  v_fma_f32 v105, v105, %[iz], %[cfz]\n"
        RTD_HYB_TAIL
        "2:\n"
        : [off] "+v"(off), [pend] "+v"(pend), [ax] "=&v"(ax), [ay] "=&v"(ay), [az] "=&v"(az), [cnt] "=&s"(cnt), [save] "=&s"(save), [save2] "=&s"(save2) RTD_TRIP_OPS
        : [end] "s"(end), [lim] "s"(lim), [thr] "s"(thr), [stop] "s"(stop), [base] "s"(base), [nx] "v"(c.nX), [ny] "v"(c.nY), [nz] "v"(c.nZ), [ix] "v"(c.ix), [iy] "v"(c.iy), [iz] "v"(c.iz),
          [cnx] "v"(c.cnx), [cny] "v"(c.cny), [cnz] "v"(c.cnz), [cfx] "v"(c.cfx), [cfy] "v"(c.cfy), [cfz] "v"(c.cfz)
        : "vcc", "scc", "memory", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109");
#undef RTD_HYB_TAIL
    return off;
}
// the oldest entry of that queue, removed from it
RTD_INLINE int pend_pop_wide(uint32_t &pend0, uint32_t &pend1) {
    const uint32_t e = pend0;
    pend0 = pend1;
    pend1 = 0u;
    return (int) ((e & RTD_PEND_WIDE) ? (e & ~RTD_PEND_WIDE) : (e & (RTD_PEND_MARK - 1u)));
}

// A leaf the single-precision filter let through (node_loop_lds32).  The reference tests the sphere iff the ray hits the leaf's own
// box (Scene.fs:41), so a sphere hit found here counts only if BoundingBox.hits(leaf box) holds -- exactly.  That test is IMPLIED by
// the sphere hit itself in all but a sliver of cases, and is evaluated only in that sliver:
//   Claim.  Let the scene satisfy  0 < r <= 100 for every bounded sphere,  |every box coordinate| <= bmax <= 1000  and
//   r_max * bmax <= 500  (SceneOffsets::box_implied, set by the host), let d be a unitised direction (|d|^2 = 1 +- 4 ulp: every ray of the render kernel),
//   let dd = |o - c|^2 <= 1e6, and let Sphere.firstIntersection return t > 1e-8 from its Greater branch (computed discriminant
//   >= 1e-8).  Then BoundingBox.hits of the box {c - r, c + r} (each corner rounded once, as Sphere.make forms it) evaluates to true.
//   Proof.  In real arithmetic on the given doubles write disc' for the true discriminant; the computed one differs by at most
//   6 ulp of (b^2 + dd + r^2) + dd * | |d|^2 - 1 | <= 2.3e-9, so disc' >= 7.7e-9.  Every computed slab distance
//   fl(fl(face - o_a) * fl(1/d_a)) is within E * |1/d_a| of the real ((c_a +- r) - o_a) / d_a with
//   E = 3.5e-16 * (|o_a - c_a| + r) + 1.2e-16 * (|c_a| + r) <= 5.05e-13 (three roundings, and the face's own rounding).
//   (A) origin on or outside the sphere: both roots are positive, so the chord's midpoint parameter tm is > 0; the midpoint lies
//   within sqrt(r^2 - disc') of the centre, i.e. inside the box by mu = r - sqrt(r^2 - disc') >= disc' / (2 r) >= 3.8e-11 on every
//   axis; hence near_a <= tm - (mu - E)|1/d_a| and far_a >= tm + (mu - E)|1/d_a| as computed, mu - E > 0: tMax >= tMin, tMax > 0
//   (also after the y stage).  (B) origin strictly inside: the roots t1 < 0 < t2 satisfy |t1| t2 = -cc / |d|^2 with t2 > 1e-8 - 1e-12
//   and |t1| >= sqrt(disc') >= 8.7e-5, so the origin is rho = r - |o - c| >= -cc / (2 r) >= 4.3e-13 / r deep, inside the box by rho on
//   every axis, and rho > 1.2e-16 * bmax (the faces' rounding) because r * bmax <= 500: every computed near distance is <= 0 (or
//   -inf, NaN for d_a = 0: ignored), every far distance > 0: tMin = 0 < tMax.  Axis-aligned directions: the differences face - o_a
//   are non-zero with the right sign by the same margins, so the products are +-inf of the right sign.  QED.
// Outside the claim's hypotheses (the Equal branch |disc| < 1e-8, a far-away origin, a scene with huge or negative radii) the
// candidate's box test is made exactly: the box is Sphere.make's (Sphere.fs:333-336), centre + (-radius) and centre + radius per axis,
// the same two IEEE additions the host made (x - (-y) = x + y and x + (-y) = x - y exactly), near/far chosen by the sign of the
// inverse direction as bbox_hits does.  Tests: every render test compares this variant with the counting one, which runs the exact
// test on every leaf; test_leaf_box_is_implied_by_the_sphere_hit aims rays at the poles and the silhouettes of spheres.
#define RTD_IMPLIED_DD 1.0e6
template <bool LDS>
RTD_INLINE void leaf_test_object_exact(const SceneView<LDS> &sc, V3 o, V3 d, double &bestF, Walk &w, int prim, bool implied) {
    const d2 g0 = sc.geo[prim * 3 + 0], g1 = sc.geo[prim * 3 + 1];
    // Sphere.firstIntersection (sphere_first_intersection above, statement for statement; dd and the branch are needed below)
    const V3 diff = vsub(o, mk(g0.x, g0.y, g1.x));
    const double b = dot(d, diff);
    const double dd = dot(diff, diff);
    const double cc = dd - g1.y;
    const double disc = (b * b - cc);
    const int cmp = fcmp(disc, 0.0);
    double t = __builtin_nan("");
    if (cmp == CMP_EQ) { const double i = (-b); t = fpos(i) ? i : t; }
    else if (cmp == CMP_GT) {
        const double s = sqrt_above_tol(disc);
        const double i1 = s - b, i2 = -(b + s);
        t = fpos(i2) ? i2 : (fpos(i1) ? i1 : t);
    }
    const double a = t * t;
    bool cand = a < bestF || (a == bestF && prim < w.best);
    if (cand && !(implied && cmp == CMP_GT && dd <= RTD_IMPLIED_DD)) { // the sliver: the leaf's BoundingBox.hits, exactly
        const double r = sc.geo[prim * 3 + 2].y;
        const double ix = 1.0 / d.x, iy = 1.0 / d.y, iz = 1.0 / d.z;
        const double rx = ix < 0.0 ? r : -r, ry = iy < 0.0 ? r : -r, rz = iz < 0.0 ? r : -r;
        cand = bbox_hits_nf(ix, iy, iz, o, g0.x + rx, g0.x - rx, g0.y + ry, g0.y - ry, g1.x + rz, g1.x - rz);
    }
    if (cand) { bestF = a; w.best = prim; w.bestLen = t; }
}
// UnboundedObjects, in array order, accepted only when Float.compare a bestFloat = Less (Scene.fs:77-86)
template <bool LDS, bool COUNT>
RTD_INLINE void unbounded_tests(const SceneView<LDS> &sc, V3 o, V3 d, Walk &w, Counters &cnt) {
    double bestF = (w.best < 0) ? __builtin_inf() : w.bestLen * w.bestLen;
    for (int u = 0; u < sc.n_unbounded; ++u) {
        int obj = sc.n_bounded + u;
        d2 g0 = sc.geo[obj * 3 + 0], g1 = sc.geo[obj * 3 + 1];
        i2 m = sc.meta[obj];
        if (COUNT) cnt.prim++;
        double t;
        if ((m.x & 3) == (int) RTD_KIND_PLANE) {
            d2 g2 = sc.geo[obj * 3 + 2];
            t = plane_intersection(o, d, mk(g0.x, g0.y, g1.x), mk(g1.y, g2.x, g2.y));
        } else t = sphere_first_intersection(o, d, mk(g0.x, g0.y, g1.x), g1.y);
        if (t == t) { // ValueSome
            double a = t * t;
            if (fcmp(a, bestF) == CMP_LT) { bestF = a; w.best = obj; w.bestLen = t; }
        }
    }
}

// The whole of Scene.hitObject on one lane (unit hooks and trace_ray; the render kernel schedules the pieces itself).
// Returns the object index or -1; `bestLen` is the ray parameter of the hit.
template <bool LDS, bool COUNT>
RTD_INLINE int hit_object(const SceneView<LDS> &sc, V3 o, V3 d, double &bestLen, Counters &cnt) {
    if (COUNT) cnt.rays++;
    Walk w;
    walk_begin(w, sc.first);
    WalkCtx c = walk_ctx(d, w);
    const int end = sc.end;
    for (;;) {
        while (w.off < end) {
            if (COUNT) cnt.aabb++;
            node_step<LDS>(sc, o, c, w);
        }
        if (!(w.off & RTD_LEAF)) break;
        if (COUNT) cnt.prim++;
        leaf_test<LDS>(sc, o, d, c, w);
    }
    unbounded_tests<LDS, COUNT>(sc, o, d, w, cnt);
    bestLen = w.bestLen;
    return w.best;
}

// ---- Textures (Texture.fs:50-67, Sphere.planeMapInverse Sphere.fs:55-61) --------------------------------------------
// `int (v * float (n - 1))` of Texture.fs:65-66 as an index into n texels.  For a coordinate in [0, 1] this is the reference's
// truncation.  Outside it -- a point a rounding error off the sphere makes Math.Acos return NaN, and `int NaN` then indexes out of
// range -- the reference throws IndexOutOfRangeException; the defined behaviour here (and in the oracle): NaN and negative
// products take texel 0, products beyond the last texel take the last one.
RTD_INLINE int texel_index(double v, int n) {
    const double t = v * (double) (n - 1);
    if (!(t >= 0.0)) return 0;
    if (t >= (double) (n - 1)) return n - 1;
    return (int) t;
}
RTD_INLINE uint32_t ramp_byte(double v) { // byte (v * 255.0) (RayTracing.App/SampleImages.fs:606-627); NaN -> 0 on both sides
    const double t = v * 255.0;
    return (t == t) ? (((uint32_t) (int32_t) t) & 0xFFu) : 0u;
}
// Rare (only textured spheres), transcendental-heavy (double-double arithmetic, rt_trig.h).  The render kernel evaluates it INLINE in a
// stage of its own (stage_tex): a call anywhere in the kernel's loop makes every value that lives across it compete for the few
// callee-saved registers (measured: 136 SGPR and 40 VGPR spills against 17 and 0 without the call, config 5 25 % slower).
// The unit hooks and the single-lane helpers call the out-of-line copy below.
RTD_INLINE uint32_t texture_colour_at_inline(const TexRec *tex, const uint8_t *texels, int id, V3 p, double *uv) {
    const TexRec root = tex[id];
    if (root.kind == 0u) return root.rgb; // ParameterisedTexture.toTexture's Colour case (Texture.fs:71)
    double inv = 1.0 / root.map_radius;
    V3 v = vscale(inv, vsub(p, mk(root.cx, root.cy, root.cz)));
    const double PI = 3.14159265358979323846;
    double theta = rtt::cr_acos(-v.y);           // Math.Acos / Math.Atan2 / Math.Sin: the correctly rounded values (rt_trig.h)
    double phi = rtt::cr_atan2(-v.z, v.x) + PI;
    double x = (phi / (2.0 * PI));
    double y = theta / PI;
    if (uv) { uv[0] = x; uv[1] = y; }
    int cur = id;
    for (int depth = 0; depth < 16; ++depth) { // Checkered trees descend to strictly smaller indices
        const TexRec t = tex[cur];
        if (t.kind == 1u) { // Checkered (Texture.fs:56-62)
            double sine = rtt::cr_sin(t.grid * x) * rtt::cr_sin(t.grid * y);
            cur = (fcmp(sine, 0.0) == CMP_LT) ? t.even : t.odd;
            continue;
        }
        if (t.kind == 2u) { // Image (Texture.fs:63-67): truncating int conversions
            int xi = texel_index(1.0 - x, t.width);
            int yi = texel_index(y, t.height);
            const uint8_t *px = texels + t.texel_off + ((size_t) yi * (size_t) t.width + (size_t) xi) * 3;
            return (uint32_t) px[0] | ((uint32_t) px[1] << 8) | ((uint32_t) px[2] << 16);
        }
        if (t.kind == 3u) { // the UV ramps of RayTracing.App/SampleImages.fs:606-627: byte (x * 255.0)
            uint32_t c = 0;
            for (int k = 0; k < 3; ++k) {
                uint32_t src = (t.ramp >> (8 * k)) & 0xFFu;
                uint32_t ch = (t.rgb >> (8 * k)) & 0xFFu;
                if (src == 1u) ch = ramp_byte(x);
                else if (src == 2u) ch = ramp_byte(y);
                c |= ch << (8 * k);
            }
            return c;
        }
        return t.rgb; // Colour
    }
    return RTD_BLACK;
}
__device__ __noinline__ uint32_t texture_colour_at(const TexRec *tex, const uint8_t *texels, int id, V3 p, double *uv) {
    return texture_colour_at_inline(tex, texels, id, p, uv);
}

// ---- Hittable.Reflection (Hittable.fs:8-12 -> Sphere.reflection Sphere.fs:150-300, InfinitePlane.reflection
//      InfinitePlane.fs:43-99) --------------------------------------------------------------------------------
// Returns true when the path is absorbed (ValueSome colour -> `colour` holds it); otherwise o/d/colour are the
// outgoing LightRay.  Stages: decide (colour, random draw, which of refract/reflect/fuzz/lambert follow), then each
// stage once, so Pure/Fuzzed/Dielectric/Glass spheres and Pure/Fuzzed planes all share one mirror computation.
enum { ACT_REFLECT = 1, ACT_REFRACT = 2, ACT_FUZZ = 4, ACT_LAMBERT = 8, ACT_LAMBERT_ONCE = 16 };

// TEX = false compiles the function without the (out-of-line) texture evaluation: for scenes that have no parameterised texture,
// and for the render kernel, which evaluates textures in a stage of their own (rt_render_kernel.h, stage_tex) and hands the colour
// over in `texPre` (RTD_NO_TEX: the object's plain colour applies).  A texture call in the middle of the general reflection cost
// the kernel its spills (128 VGPRs, ~50 VGPR and ~115 SGPR spills, 204 B of scratch) and ran the whole double-double evaluation for
// the one or two textured lanes of almost every batch.
#define RTD_NO_TEX 0xFFFFFFFFu
// does the hit object's colour come from a parameterised texture?  Styles that carry a Texture: every SphereStyle but LightSourceCap
// (Sphere.fs:10-37), and the plane LightSource (InfinitePlane.fs:5); the other plane styles carry a plain Pixel (the host rejects a
// texture id elsewhere).  Returns the texture's id or -1.
RTD_INLINE int textured(i2 m) {
    const int texId = (int) (((uint32_t) m.y) >> 24) - 1;
    const bool isPlane = (m.x & 3) == (int) RTD_KIND_PLANE;
    const int style = (m.x >> 2) & 7;
    return (texId >= 0 && (isPlane ? style == 0 : style != 1)) ? texId : -1;
}
template <bool LDS, bool TEX = true>
RTD_INLINE bool reflection(const SceneView<LDS> &sc, int obj, V3 strike, V3 &o, V3 &d, uint32_t &colour, Rng &rng, uint32_t texPre = RTD_NO_TEX) {
    const i2 m = sc.meta[obj];
    const bool isPlane = (m.x & 3) == (int) RTD_KIND_PLANE;
    const int style = (m.x >> 2) & 7;
    const d2 g0 = sc.geo[obj * 3 + 0], g1 = sc.geo[obj * 3 + 1], g2 = sc.geo[obj * 3 + 2];
    const double m0 = sc.mat[obj * 4 + 0], p1 = sc.mat[obj * 4 + 1], p2 = sc.mat[obj * 4 + 2], p3 = sc.mat[obj * 4 + 3];
    const double albedo = isPlane ? m0 : g2.x; // a sphere's albedo rides in its geo record (the common Lambert case never reads `mat`)
    uint32_t texColour = (uint32_t) m.y & 0x00FFFFFFu;
    if (texPre != RTD_NO_TEX) texColour = texPre;
    else if (TEX) {
        const int texId = textured(m);
        if (texId >= 0) texColour = texture_colour_at(sc.tex, sc.texels, texId, strike, nullptr);
    }

    V3 n;              // normal.Vector (possibly flipped)
    bool inside = false;
    int act = 0;
    double cosI = 0.0; // incomingCos handed to refract
    double ior = p1, fuzz = p1;
    double invIor = p3; // 1.0 / ior, formed once per material by the host (rt_scene.h) exactly as Sphere.fs:117,284 form it per hit

    if (isPlane) {
        n = mk(g1.y, g2.x, g2.y);
        // InfinitePlaneStyle order: LightSource, PureReflection, LambertReflection, FuzzedReflection
        if (style == 0) { colour = pix_combine(colour, texColour); return true; } // InfinitePlane.fs:52-56
        colour = pix_darken(albedo, pix_combine(colour, texColour));              // newColour, InfinitePlane.fs:40-41
        if (style == 1) act = ACT_REFLECT;
        else if (style == 3) act = ACT_REFLECT | ACT_FUZZ;
        else act = ACT_LAMBERT_ONCE;
    } else {
        const V3 c = mk(g0.x, g0.y, g1.x);
        const double r2 = g1.y, radius = g2.y;
        const bool flipped = (m.x >> 5) & 1; // Float.compare radius 0.0 = Less (Sphere.fs:321), set by the host
        if (!unitise(vsub(strike, c), n)) n = mk(0.0, 0.0, 0.0); // Sphere.normal (Sphere.fs:65-66)
        V3 co = vsub(c, o);
        int cmp = fcmp(dot(co, co), r2); // Sphere.fs:165-179
        if ((cmp != CMP_GT) != flipped) { inside = true; n = vscale(-1.0, n); }

        if (style == 0) { colour = pix_combine(colour, texColour); return true; } // LightSource, Sphere.fs:185-189
        if (style == 1) {                                                         // LightSourceCap, Sphere.fs:190-200
            double lower = c.x + (radius - (radius / 4.0));
            colour = (fcmp(strike.x, lower) == CMP_GT) ? pix_combine(texColour, colour) : RTD_BLACK;
            return true;
        }
        colour = pix_darken(albedo, pix_combine(colour, texColour)); // every remaining style: combine then darken
        if (style == 4) act = ACT_LAMBERT;                           // Sphere.fs:202-222
        else if (style == 2) act = ACT_REFLECT;                      // Sphere.fs:224-233
        else if (style == 3) act = ACT_REFLECT | ACT_FUZZ;           // Sphere.fs:235-246
        else if (style == 5) {                                       // Dielectric, Sphere.fs:248-267
            double r = rng_get(rng);
            if (r > p2) act = ACT_REFLECT;
            else { cosI = dot(d, n); act = ACT_REFRACT; }
        } else {                                                     // Glass, Sphere.fs:269-300
            cosI = dot(vscale(-1.0, d), n);
            double r = rng_get(rng);
            invIor = p2;
            const double param = inside ? m0 : p3; // ((1 - sr) / (1 + sr))^2 for sr = 1/ior | ior (Sphere.fs:283-289), per material
            double prob = param + (1.0 - param) * pow5(1.0 - cosI);
            act = (r < prob) ? ACT_REFLECT : ACT_REFRACT;
        }
    }

    // Plane.makeOrthonormalSpannedBy normal ray (Plane.fs:64-79): shared by refract and the mirror
    V3 v2 = mk(0.0, 0.0, 0.0);
    bool haveV2 = false;
    if (act & (ACT_REFLECT | ACT_REFRACT)) {
        double coefficient = dot(n, d);
        haveV2 = unitise(vsub(d, vscale(coefficient, n)), v2);
    }

    // The three geometric stages below each end in "unitise (target - strike) and overwrite the ray"; they only PREPARE
    // the vector, one shared normalisation follows.  `originToStrike` covers the one asymmetry: a failed overwriteWithMake
    // leaves a sphere's ray untouched (Ray.fs:14-15), whereas the plane's pureOutgoing builds a new ray at the strike point.
    V3 nv = mk(0.0, 0.0, 0.0);
    bool haveNv = false, originToStrike = false;

    if (act & ACT_REFRACT) { // Sphere.refract (Sphere.fs:108-146)
        double index = inside ? invIor : ior; // 1.0 / index | index / 1.0 (Sphere.fs:117)
        if (!haveV2) { nv = d; haveNv = true; } // parallel to the normal: straight through, re-normalised (Sphere.fs:121-124)
        else {
            double sinI = sqrt(1.0 - cosI * cosI);
            double sinO = sinI / index;
            if (fcmp(sinO, 1.0) == CMP_GT) act |= ACT_REFLECT; // total internal reflection (Sphere.fs:130-132)
            else {
                double cosO = sqrt(1.0 - sinO * sinO);
                V3 outP = walk(walk(strike, n, (-cosO)), v2, sinO);
                nv = vsub(outP, strike);
                haveNv = true;
            }
        }
    }

    if (act & ACT_REFLECT) { // Sphere.reflectWithoutFuzz (Sphere.fs:68-87) / InfinitePlane.pureOutgoing (InfinitePlane.fs:18-38)
        if (!haveV2) { // directly along the normal: flip
            d = vscale(-1.0, d);
            o = strike;
        } else {
            double nC = -dot(n, d);
            double tC = dot(v2, d);
            V3 dest = walk(walk(strike, n, nC), v2, tC);
            nv = vsub(dest, strike);
            haveNv = true;
            originToStrike = isPlane;
        }
    }

    if (haveNv) {
        V3 nd;
        bool ok = unitise(nv, nd);
        if (ok) d = nd;
        if (ok || originToStrike) o = strike;
    }

    // Sphere.addFuzz (Sphere.fs:89-104) / InfinitePlane.fs:63-71: unit offset scaled by fuzz around (ray.origin + ray.dir);
    // Lambert, Sphere.fs:211-220 (retry) / InfinitePlane.fs:79-86 (single try): unit offset around (strike + normal).
    // Same arithmetic shape -- centre = base + dir*1.0, target = centre + offset*scale -- so one loop serves both.
    if (act & (ACT_FUZZ | ACT_LAMBERT | ACT_LAMBERT_ONCE)) {
        const bool fz = (act & ACT_FUZZ) != 0;
        const V3 centre = fz ? walk(o, d, 1.0) : walk(strike, n, 1.0);
        const double scale = fz ? fuzz : 1.0;
        for (;;) {
            V3 offset = random_unit(rng);
            V3 target = walk(centre, offset, scale);
            V3 nd;
            if (unitise(vsub(target, strike), nd)) { o = strike; d = nd; break; }
            if (act & ACT_LAMBERT_ONCE) { colour = RTD_BLACK; return true; } // the reference throws here; see DESIGN.md
        }
    }
    return false;
}

// The two cases that make up ~85 % of all path vertices of the reference's scenes, as a function of their own: a light
// source without a parameterised texture (Sphere.fs:185-189, InfinitePlane.fs:52-56) and an untextured LambertReflection
// sphere (Sphere.fs:162-182, 202-222).  Statement for statement what `reflection` executes for those objects -- the render
// kernel shades them where they fall and hands every other style to `reflection` in batches (rt_render_kernel.h).
// `fast_style` says which objects qualify (from the meta word alone).
RTD_INLINE bool fast_style(i2 m) {
    const uint32_t ks = (uint32_t) m.x & 31u; // kind | style << 2
#ifdef RTD_PARK_LAMBERT /* experiment (VERDICT r2 item 1's gate): park everything but the light sources */
    return (((uint32_t) m.y) >> 24) == 0u && (ks == (RTD_KIND_SPHERE | (0u << 2)) || ks == (RTD_KIND_PLANE | (0u << 2)));
#endif
    return (((uint32_t) m.y) >> 24) == 0u && (ks == (RTD_KIND_SPHERE | (0u << 2)) || ks == (RTD_KIND_SPHERE | (4u << 2)) || ks == (RTD_KIND_PLANE | (0u << 2)));
}
// The Lambert case in two pieces, so that the render kernel can park a Lambert hit between them as {strike, inside} instead of the
// whole incoming ray: Sphere.reflection's prologue (Sphere.fs:162-182: is the ray's origin inside the sphere?) ...
template <bool LDS>
RTD_INLINE bool lambert_inside(const SceneView<LDS> &sc, int obj, i2 m, V3 o) {
    const d2 g0 = sc.geo[obj * 3 + 0], g1 = sc.geo[obj * 3 + 1];
    const bool flipped = (m.x >> 5) & 1;
    const V3 co = vsub(mk(g0.x, g0.y, g1.x), o);
    const int cmp = fcmp(dot(co, co), g1.y); // Sphere.fs:165-179
    return (cmp != CMP_GT) != flipped;
}
// ... and the bounce itself from the strike point (Sphere.normal Sphere.fs:65-66, LambertReflection Sphere.fs:202-222)
template <bool LDS>
RTD_INLINE void lambert_bounce(const SceneView<LDS> &sc, int obj, i2 m, V3 strike, bool inside, V3 &o, V3 &d, uint32_t &colour, Rng &rng) {
    const uint32_t texColour = (uint32_t) m.y & 0x00FFFFFFu;
    const d2 g0 = sc.geo[obj * 3 + 0], g1 = sc.geo[obj * 3 + 1];
    const double albedo = sc.geo[obj * 3 + 2].x;
    V3 n;
    if (!unitise(vsub(strike, mk(g0.x, g0.y, g1.x)), n)) n = mk(0.0, 0.0, 0.0); // Sphere.normal (Sphere.fs:65-66)
    if (inside) n = vscale(-1.0, n);
    colour = pix_darken(albedo, pix_combine(colour, texColour)); // Sphere.fs:203-207
    const V3 centre = walk(strike, n, 1.0);                       // Sphere.fs:211-220
    for (;;) {
        V3 offset = random_unit(rng);
        V3 target = walk(centre, offset, 1.0);
        V3 nd;
        if (unitise(vsub(target, strike), nd)) { o = strike; d = nd; break; }
    }
}
template <bool LDS>
RTD_INLINE bool reflection_fast(const SceneView<LDS> &sc, int obj, i2 m, V3 strike, V3 &o, V3 &d, uint32_t &colour, Rng &rng) {
    const uint32_t texColour = (uint32_t) m.y & 0x00FFFFFFu;
    if ((((uint32_t) m.x >> 2) & 7u) == 0u) { colour = pix_combine(colour, texColour); return true; } // LightSource, sphere or plane
    const bool inside = lambert_inside<LDS>(sc, obj, m, o);
    lambert_bounce<LDS>(sc, obj, m, strike, inside, o, d, colour, rng);
    return false;
}

// ---- Scene.traceOnce's ray construction (Scene.fs:129-144) ---------------------------------------------------------
struct CameraParams {
    double eye[3], xo[3], xd[3], yd[3];
    double vw, vh;
    int32_t max_w, max_h;
    int32_t spp, depth;
};
RTD_INLINE bool camera_ray(const CameraParams &cam, int row, int col, Rng &rng, V3 &o, V3 &d) {
    double r1 = rng_get(rng), r2 = rng_get(rng); // GetTwo
    double landing = (((double) col + r1) * cam.vw) / (double) cam.max_w;
    V3 onX = walk(mk(cam.xo[0], cam.xo[1], cam.xo[2]), mk(cam.xd[0], cam.xd[1], cam.xd[2]), landing);
    double wd = (((double) row + r2) * cam.vh) / (double) cam.max_h;
    V3 end = walk(onX, mk(cam.yd[0], cam.yd[1], cam.yd[2]), wd);
    o = mk(cam.eye[0], cam.eye[1], cam.eye[2]);
    return unitise(vsub(end, o), d);
}

// ---- the leaves a pixel's camera rays can reach, found ONCE per pixel -----------------------------------------------------------
// Every sample of a pixel starts with a ray from the eye through the pixel's patch of the viewport (Scene.fs:129-144: the patch is
// the parallelogram {xo + lx * xd + ly * yd : lx in [col, col+1] * vw / maxW, ly in [row, row+1] * vh / maxH}), and with hundreds of
// samples per pixel those rays -- a third of all rays of the bench frame -- walk the tree to the same few leaves again and again.
// pixel_candidates walks the tree ONCE with the whole pyramid of the pixel's rays (apex at the eye, through the patch's four corners)
// and returns the Leaves whose boxes the pyramid can touch, as the node loop's queue word (rt_device.h, node_loop_lds32); a camera
// ray of that pixel then starts with its tree walk already exhausted and those Leaves in its queue, and the leaf pass makes each
// Leaf's exact BoundingBox.hits (or knows it implied) and sphere test as for any other queued Leaf.  The result is the reference's:
// a sphere is tested iff its Leaf box is hit by the ray, exactly -- the candidates only have to CONTAIN every Leaf the ray's exact
// test could accept.  They do: box and pyramid are convex, so if all of the box lies strictly outside one of the pyramid's four side
// planes (or behind the eye), no ray of the pixel meets it; the test below rejects a box only then.  The planes are set up in double
// precision (rounding ~1e-12 relative: the cross products cancel to ~1e-4 of their operands for a 2401-pixel-wide image) and then
// held, like the box's centre and half extents, in single precision; the per-box test runs in single precision with a margin of
// 2^-18 of its terms' magnitudes (every operand carries 2^-24, the sums a few times that; a ray that the exact slab test accepts
// misses the box by at most a few double-precision ulps of its coordinates).  The box tested is the node32 record's (rounded
// outward: larger).  More than FOUR reachable Leaves, a degenerate pyramid, or a kernel variant without the queue: RTD_CAND_WALK,
// and the pixel's camera rays walk the tree as all others.
// The counting kernel variant never uses candidates, and every render test compares the two variants and the oracle.
#define RTD_CAND_WALK 0xFFFFFFFFu
struct F3 { float x, y, z; };
RTD_INLINE V3 cross3(V3 a, V3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
RTD_INLINE F3 to_f3(V3 v) { F3 r; r.x = (float) v.x; r.y = (float) v.y; r.z = (float) v.z; return r; }
// the whole box (centre c relative to the eye, half extents h >= 0) lies on the negative side of the plane n . x = 0.  Single
// precision: n, c, h carry 2^-24 relative rounding each, the sums a few more; the margin is 2^-18 of the terms' magnitudes.
RTD_INLINE bool outside_plane(F3 n, F3 c, F3 h) {
    const float ax = __builtin_fabsf(n.x), ay = __builtin_fabsf(n.y), az = __builtin_fabsf(n.z);
    const float s = ((n.x * c.x + n.y * c.y) + n.z * c.z) + ((ax * h.x + ay * h.y) + az * h.z);
    const float m = 0x1p-18f * ((ax * (__builtin_fabsf(c.x) + h.x) + ay * (__builtin_fabsf(c.y) + h.y)) + az * (__builtin_fabsf(c.z) + h.z));
    return s < -m; // (a NaN anywhere compares false: not rejected)
}
// Returns up to four reachable Leaves as two queue words (the older pair in `first`), RTD_CAND_WALK in `first` for "walk the tree".
template <bool LDS, bool USE> RTD_INLINE uint32_t pixel_candidates(const SceneView<LDS> &sc, const CameraParams &cam, int row, int col, uint32_t &second) {
  second = 0u;
  if constexpr (!USE) return RTD_CAND_WALK; // only the timed variants have the queue the candidates go into
  else {
    const V3 eye = mk(cam.eye[0], cam.eye[1], cam.eye[2]);
    F3 n[4], gcf;
    {   // the pyramid's planes, set up in double precision (the cross products cancel to ~1e-4 of their operands) and kept in single
        const V3 xo = mk(cam.xo[0], cam.xo[1], cam.xo[2]), xd = mk(cam.xd[0], cam.xd[1], cam.xd[2]), yd = mk(cam.yd[0], cam.yd[1], cam.yd[2]);
        V3 g[4]; // corners (lx0,ly0) (lx1,ly0) (lx1,ly1) (lx0,ly1) of the patch, from the eye: camera_ray's arithmetic with r1, r2 in {0, 1}
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int jx = (q == 1 || q == 2) ? 1 : 0, jy = q >> 1;
            const double lx = (((double) col + (double) jx) * cam.vw) / (double) cam.max_w;
            const double ly = (((double) row + (double) jy) * cam.vh) / (double) cam.max_h;
            g[q] = vsub(walk(walk(xo, xd, lx), yd, ly), eye);
        }
        const V3 gc = mk((g[0].x + g[1].x) + (g[2].x + g[3].x), (g[0].y + g[1].y) + (g[2].y + g[3].y), (g[0].z + g[1].z) + (g[2].z + g[3].z));
        gcf = to_f3(gc);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            V3 nq = cross3(g[q], g[(q + 1) & 3]);
            const double sgn = dot(nq, gc);
            if (!(sgn != 0.0)) return RTD_CAND_WALK; // degenerate pyramid (or NaN): no claim
            // normalised to ~1 before rounding to single (|n| ~ 1e-2 |g|^2 otherwise: no under- or overflow for any camera)
            const double sc1 = (sgn < 0.0 ? -1.0 : 1.0) / (fabs(nq.x) + fabs(nq.y) + fabs(nq.z));
            n[q] = to_f3(vscale(sc1, nq));
        }
    }
    uint32_t pend = 0u;
    int count = 0;
    for (int off = sc.first; off < sc.end;) {
        const typename Ptrs<LDS>::bp rec = node_at<LDS>(sc, off);
        typedef typename std::conditional<LDS, const RTD_AS3 float *, const float *>::type fp;
        typedef typename std::conditional<LDS, const RTD_AS3 i4 *, const i4 *>::type i4p;
        const fp f = (fp) rec;
        const i4 lk = *(i4p) (rec + 48); // on_hit, on_miss, queue entry (0 for a Branch), shift
        const float lx = f[0], hx = f[1], ly = f[4], hy = f[5], lz = f[8], hz = f[9];
        F3 c, h;
        c.x = (float) ((0.5 * ((double) lx + (double) hx)) - eye.x); c.y = (float) ((0.5 * ((double) ly + (double) hy)) - eye.y); c.z = (float) ((0.5 * ((double) lz + (double) hz)) - eye.z);
        h.x = 0.5f * __builtin_fabsf(hx - lx) + 1e-30f; h.y = 0.5f * __builtin_fabsf(hy - ly) + 1e-30f; h.z = 0.5f * __builtin_fabsf(hz - lz) + 1e-30f;
        h.x += 0x1p-22f * (__builtin_fabsf(lx) + __builtin_fabsf(hx)); h.y += 0x1p-22f * (__builtin_fabsf(ly) + __builtin_fabsf(hy)); h.z += 0x1p-22f * (__builtin_fabsf(lz) + __builtin_fabsf(hz)); // hx - lx is rounded
        const bool miss = outside_plane(n[0], c, h) || outside_plane(n[1], c, h) || outside_plane(n[2], c, h) || outside_plane(n[3], c, h) || outside_plane(gcf, c, h);
        if (!miss && lk.z != 0) {
            if (LDS || sc.narrow) { // two 16-bit entries per word (node_loop_lds32's / node_loop_hyb16's queue), two words
                if (count == 4) return RTD_CAND_WALK;
                if (count == 2) { second = pend; pend = 0u; } // the first pair is complete: it becomes the older word
                pend = (pend >> 16) | ((uint32_t) lk.z << 16); // the queue's own push: the newer entry in the high half
            } else { // full-width entries (node_loop_glb32's queue): the older in `first`, the newer in `second`
                if (count == 2) return RTD_CAND_WALK;
                if (count == 0) pend = (uint32_t) lk.z; else second = (uint32_t) lk.z;
            }
            ++count;
        }
        off = miss ? lk.y : lk.x;
    }
    if ((LDS || sc.narrow) && count > 2) { const uint32_t t = second; second = pend; pend = t; } // first = entries 1-2, second = entries 3-4
    return pend;
  }
}

// ---- Scene.traceRay (Scene.fs:93-114), whole path on one lane (unit hooks; the render kernel interleaves bounces) -------
template <bool LDS, bool COUNT>
RTD_INLINE uint32_t trace_ray(const SceneView<LDS> &sc, int maxCount, V3 o, V3 d, Rng &rng, Counters &cnt) {
    uint32_t colour = RTD_WHITE;
    int bounces = 0;
    while (bounces <= maxCount) {
        double t;
        int obj = hit_object<LDS, COUNT>(sc, o, d, t, cnt);
        if (obj < 0) return RTD_BLACK;
        V3 strike = walk(o, d, t);
        if (COUNT) cnt.refl++;
        if (reflection<LDS>(sc, obj, strike, o, d, colour, rng)) return colour;
        bounces = bounces + 1;
    }
    return RTD_HOTPINK;
}

} // namespace rtd
