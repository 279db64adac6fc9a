// oracle.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// A CPU restatement of the per-pixel sampling path of Smaug123/ray-tracing-fsharp, written to follow the
// reference's F# source literally (same operation order, same tolerance compares, same byte arithmetic,
// recursive tree walk, pointer-based BoundingBoxTree), so that the HIP path can be checked bit-for-bit.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product
// (ray-tracing-fsharp_amd/) never links, imports or calls it.
//
// Pinning: the reference is F#/.NET and cannot be built or run here (no dotnet/mono), and it
// is non-deterministic by construction (unseeded System.Random, FloatProducers shared across row tasks).
// The oracle is therefore pinned by the reference's OWN unit-test vectors (RayTracing.Test/*.fs; replayed in
// tests/test_oracle_reference_kats.py) for the functions those tests cover: Sphere.firstIntersection,
// Sphere.reflection (Glass/Dielectric), Sphere.planeMap/planeMapInverse, BoundingBox.hits, Pixel.combine,
// FloatProducer range/decile/distinctness, Ray.walkAlong, Plane.basis, ImageOutput.writePpm golden file.
// PARITY UNPINNED (no reference test or golden output exists): Scene.* (tree walk, traceRay, adaptive
// sampling), BoundingBoxTree.make, InfinitePlane.*, Lambert/Pure/Fuzzed/LightSourceCap reflection,
// Pixel.darken, PixelStats, Camera.makeBasic, textures, gamma.  Those follow the source text below.
//
// Every function cites the reference file:line it restates (paths relative to /root/reference/RayTracing).
//
// Build: g++ -O2 -std=c++17 -ffp-contract=off -fno-fast-math (RyuJIT never fuses mul+add; F7/F8 in SURVEY.md).
// libm: Math.Acos/Atan2/Sin on Linux .NET call the C runtime, i.e. glibc -- the same functions used here (textures only).
// Math.Pow(x, 5.0) is taken as the correctly rounded power (see SphereM::pow5).

#include "../include/rtfs_amd.h"

#include <atomic>
#include <cmath>
#include <quadmath.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <memory>
#include <string>
#include <thread>
#include <vector>
#include <algorithm>
#include <chrono>

namespace orc {

// ------------------------------------------------------------------------------------------------
// Float.fs
// ------------------------------------------------------------------------------------------------
enum class Comparison { Greater, Equal, Less }; // Float.fs:6-10

struct FloatProducer { // Float.fs:31-76 (the Monitor lock is dropped: one producer per path here)
    uint32_t x, y, z, w;

    // Float.fs:14-20
    static inline uint32_t generateInt32(uint32_t &x, uint32_t &y, uint32_t &z, uint32_t &w) {
        uint32_t t = x ^ (x << 11);
        x = y;
        y = z;
        z = w;
        w = w ^ (w >> 19) ^ (t ^ (t >> 8));
        return w;
    }
    // Float.fs:22-27
    static inline uint32_t toInt(uint32_t w) {
        uint32_t highest = (w & 0xFFu);
        uint32_t secondHighest = ((w >> 8) & 0xFFu);
        uint32_t thirdHighest = ((w >> 16) & 0xFFu);
        uint32_t lowest = ((w >> 24) & 0xFFu);
        return ((highest << 24) ^ (secondHighest << 16) ^ (thirdHighest << 8) ^ lowest);
    }
    // Float.fs:29
    static inline double toDouble(uint32_t i) { return (double) i / (double) 4294967295u; }

    double Get() { // Float.fs:38-47
        uint32_t w1 = generateInt32(x, y, z, w);
        return toDouble(toInt(w1));
    }
    void GetTwo(double &a, double &b) { // Float.fs:49-60
        uint32_t one = generateInt32(x, y, z, w);
        uint32_t two = generateInt32(x, y, z, w);
        a = toDouble(toInt(one));
        b = toDouble(toInt(two));
    }
    void GetThree(double &a, double &b, double &c) { // Float.fs:62-76
        uint32_t one = generateInt32(x, y, z, w);
        uint32_t two = generateInt32(x, y, z, w);
        uint32_t three = generateInt32(x, y, z, w);
        a = toDouble(toInt(one));
        b = toDouble(toInt(two));
        c = toDouble(toInt(three));
    }
};

namespace Float {
static const double tolerance = 0.00000001;                                      // Float.fs:82
static inline bool equal(double a, double b) { return std::fabs(a - b) < tolerance; } // Float.fs:84
static inline bool positive(double a) { return a > tolerance; }                  // Float.fs:88
static inline Comparison compare(double a, double b) {                           // Float.fs:90-96
    if (std::fabs(a - b) < tolerance) return Comparison::Equal;
    else if (a < b) return Comparison::Less;
    else return Comparison::Greater;
}
} // namespace Float

// ------------------------------------------------------------------------------------------------
// Stream seeding -- the one thing the build DEFINES (the reference seeds from unseeded System.Random,
// Float.fs:33-36, Scene.fs:205).  Spec in DESIGN.md "Seeding"; restated independently in the HIP path.
// ------------------------------------------------------------------------------------------------
static inline uint64_t mix64(uint64_t z) { // SplitMix64 finaliser
    z ^= z >> 30;
    z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27;
    z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}
static inline FloatProducer streamFor(uint64_t seed, uint64_t pixel, uint32_t sample) {
    const uint64_t G = 0x9E3779B97F4A7C15ull;
    uint64_t k = mix64(seed + G);                                                   // seed key
    uint64_t hp = mix64(k ^ (pixel * 0xD1B54A32D192ED03ull + 0x8CB92BA72F3D8DD7ull)); // the pixel's SplitMix64 state
    uint64_t a = mix64(hp + (2ull * sample + 1ull) * G);                            // its outputs 2s+1 and 2s+2
    uint64_t b = mix64(hp + (2ull * sample + 2ull) * G);
    FloatProducer p;
    // uint (rand.Next ()) lies in [0, 2^31-2] (Float.fs:33-36): keep the state inside the reference's domain.
    p.x = (uint32_t) (a & 0xFFFFFFFFull) % 2147483647u;
    p.y = (uint32_t) (a >> 32) % 2147483647u;
    p.z = (uint32_t) (b & 0xFFFFFFFFull) % 2147483647u;
    p.w = (uint32_t) (b >> 32) % 2147483647u;
    if ((p.x | p.y | p.z | p.w) == 0u) p.w = 1u; // xorshift's absorbing state
    return p;
}

// ------------------------------------------------------------------------------------------------
// Point.fs
// ------------------------------------------------------------------------------------------------
struct Point { double x, y, z; };  // Point.fs:7-8
struct Vector { double x, y, z; }; // Point.fs:10-11
typedef Vector UnitVector;         // Point.fs:13-14 (a wrapper with no arithmetic of its own)

namespace Vec {
static inline double dot(Vector p, Vector q) { return p.x * q.x + p.y * q.y + p.z * q.z; }         // Point.fs:18
static inline Vector sum(Vector p, Vector q) { return Vector{p.x + q.x, p.y + q.y, p.z + q.z}; }  // Point.fs:20
static inline Vector scale(double s, Vector v) { return Vector{s * v.x, s * v.y, s * v.z}; }       // Point.fs:22-24
static inline Vector difference(Vector p, Vector q) { return Vector{p.x - q.x, p.y - q.y, p.z - q.z}; } // Point.fs:26
static inline bool unitise(Vector vec, UnitVector &out) {                                           // Point.fs:28-35
    double d = dot(vec, vec);
    if (Float::equal(d, 0.0)) return false;
    double factor = 1.0 / std::sqrt(d);
    out = scale(factor, vec);
    return true;
}
static inline double normSquared(Vector v) { return dot(v, v); }                                    // Point.fs:37
static inline bool equal(Vector p, Vector q) {                                                      // Point.fs:39-40
    return Float::equal(p.x, q.x) && Float::equal(p.y, q.y) && Float::equal(p.z, q.z);
}
static inline Vector make(double x, double y, double z) { return Vector{x, y, z}; }                 // Point.fs:42
static inline Vector cross(Vector p, Vector q) {                                                    // Point.fs:44-45
    return make(p.y * q.z - p.z * q.y, p.z * q.x - p.x * q.z, p.x * q.y - q.x * p.y);
}
} // namespace Vec

namespace UV {
static UnitVector random(FloatProducer &floatProducer) { // Point.fs:49-59 (recursion written as a loop)
    for (;;) {
        double rand1, rand2, rand3;
        floatProducer.GetThree(rand1, rand2, rand3);
        double x = (2.0 * rand1) - 1.0;
        double y = (2.0 * rand2) - 1.0;
        double z = (2.0 * rand3) - 1.0;
        UnitVector result;
        if (Vec::unitise(Vec::make(x, y, z), result)) return result;
    }
}
static inline UnitVector flip(UnitVector v) { return Vec::scale(-1.0, v); } // Point.fs:66
static inline double coordinate(int i, UnitVector v) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); } // Point.fs:75-80
} // namespace UV

namespace Pt {
static inline double coordinate(int i, Point p) { return i == 0 ? p.x : (i == 1 ? p.y : p.z); }      // Point.fs:85-90
static inline Point sum(Point p, Point q) { return Point{p.x + q.x, p.y + q.y, p.z + q.z}; }         // Point.fs:92
static inline Vector differenceToThenFrom(Point p, Point q) { return Vector{p.x - q.x, p.y - q.y, p.z - q.z}; } // Point.fs:94
static inline bool equal(Point p, Point q) {                                                          // Point.fs:96-97
    return Float::equal(p.x, q.x) && Float::equal(p.y, q.y) && Float::equal(p.z, q.z);
}
static inline Point make(double x, double y, double z) { return Point{x, y, z}; }                    // Point.fs:99
} // namespace Pt

// ------------------------------------------------------------------------------------------------
// Pixel.fs
// ------------------------------------------------------------------------------------------------
struct Pixel { uint8_t Red, Green, Blue; }; // Pixel.fs:9-15

namespace Colour {
static const Pixel Black{0, 0, 0};         // Pixel.fs:19-24
static const Pixel White{255, 255, 255};   // Pixel.fs:26-31
static const Pixel HotPink{205, 105, 180}; // Pixel.fs:61-66
} // namespace Colour

struct PixelStats { int Count, SumRed, SumGreen, SumBlue; }; // Pixel.fs:78-85

namespace PixelStatsM {
static inline PixelStats empty() { return PixelStats{0, 0, 0, 0}; } // Pixel.fs:89-95
static inline void add(Pixel p, PixelStats &stats) {                // Pixel.fs:97-101
    stats.Count = stats.Count + 1;
    stats.SumRed = stats.SumRed + (int) p.Red;
    stats.SumGreen = stats.SumGreen + (int) p.Green;
    stats.SumBlue = stats.SumBlue + (int) p.Blue;
}
static inline Pixel mean(const PixelStats &stats) {                 // Pixel.fs:103-108
    return Pixel{(uint8_t) (stats.SumRed / stats.Count), (uint8_t) (stats.SumGreen / stats.Count),
                 (uint8_t) (stats.SumBlue / stats.Count)};
}
} // namespace PixelStatsM

namespace PixelM {
static inline int difference(Pixel p1, Pixel p2) { // Pixel.fs:113-116
    return std::abs((int) p1.Red - (int) p2.Red) + std::abs((int) p1.Green - (int) p2.Green) +
           std::abs((int) p1.Blue - (int) p2.Blue);
}
static inline Pixel combine(Pixel p1, Pixel p2) { // Pixel.fs:136-141
    return Pixel{(uint8_t) (((int) p1.Red * (int) p2.Red) / 255), (uint8_t) (((int) p1.Green * (int) p2.Green) / 255),
                 (uint8_t) (((int) p1.Blue * (int) p2.Blue) / 255)};
}
// Math.Round is round-half-to-even = rint() in the default rounding mode.
static inline uint8_t roundByte(double v) { return (uint8_t) (int32_t) std::rint(v); }
static inline Pixel darken(double albedo, Pixel p) { // Pixel.fs:144-151
    return Pixel{roundByte((double) p.Red * albedo), roundByte((double) p.Green * albedo),
                 roundByte((double) p.Blue * albedo)};
}
} // namespace PixelM

// ------------------------------------------------------------------------------------------------
// Ray.fs
// ------------------------------------------------------------------------------------------------
struct Ray { Point Origin; UnitVector Vector; }; // Ray.fs:3-7

namespace RayM {
static inline bool overwriteWithMake(Point origin, Vector vector, Ray &ray) { // Ray.fs:11-24
    double d = Vec::dot(vector, vector);
    if (Float::equal(d, 0.0)) return false;
    ray.Origin = origin;
    double factor = 1.0 / std::sqrt(d);
    ray.Vector = Vec::scale(factor, vector);
    return true;
}
static inline bool makePrime(Point origin, Vector vector, Ray &out) { // Ray.fs:26-34 (make')
    UnitVector v;
    if (!Vec::unitise(vector, v)) return false;
    out.Origin = origin;
    out.Vector = v;
    return true;
}
static inline Ray make(Point origin, UnitVector vector) { return Ray{origin, vector}; } // Ray.fs:36-40
static inline Point walkAlongRay(Point o, UnitVector v, double magnitude) {            // Ray.fs:42-43
    return Pt::make(o.x + (v.x * magnitude), o.y + (v.y * magnitude), o.z + (v.z * magnitude));
}
static inline Point walkAlong(const Ray &ray, double magnitude) { return walkAlongRay(ray.Origin, ray.Vector, magnitude); } // Ray.fs:45-46
static inline Ray parallelTo(Point p1, const Ray &ray) { return Ray{p1, ray.Vector}; }  // Ray.fs:48-52
static inline void translateToIntersect(Point p1, Ray &ray) { ray.Origin = p1; }       // Ray.fs:54
static inline bool liesOn(Point point, const Ray &ray) {                                // Ray.fs:56-66
    double t = (point.x - ray.Origin.x) / ray.Vector.x;
    double t2 = (point.y - ray.Origin.y) / ray.Vector.y;
    if (Float::equal(t, t2)) {
        double t3 = (point.z - ray.Origin.z) / ray.Vector.z;
        return Float::equal(t, t3);
    }
    return false;
}
static inline Ray flip(const Ray &r) { return Ray{r.Origin, Vec::scale(-1.0, r.Vector)}; } // Ray.fs:71-77
static inline void flipInPlace(Ray &r) { r.Vector = Vec::scale(-1.0, r.Vector); }          // Ray.fs:79-81
} // namespace RayM

// ------------------------------------------------------------------------------------------------
// Plane.fs
// ------------------------------------------------------------------------------------------------
struct OrthonormalPlane { UnitVector V1, V2; Point P; }; // Plane.fs:12-17

namespace PlaneM {
static inline OrthonormalPlane makeNormalTo(Point point, Vector v) { // Plane.fs:22-36 (ValueOption.get: asserted)
    Vector v1;
    if (Float::equal(v.z, 0.0)) v1 = Vec::make(0.0, 0.0, 1.0);
    else v1 = Vec::make(1.0, 1.0, ((-v.x - v.y) / v.z));
    UnitVector v2{0, 0, 0}, v1u{0, 0, 0};
    Vec::unitise(Vec::cross(v, v1), v2);
    Vec::unitise(v1, v1u);
    return OrthonormalPlane{v1u, v2, point};
}
static inline bool makeOrthonormalSpannedBy(const Ray &r1, const Ray &r2, OrthonormalPlane &out) { // Plane.fs:64-79
    double coefficient = Vec::dot(r1.Vector, r2.Vector);
    UnitVector v2;
    if (!Vec::unitise(Vec::difference(r2.Vector, Vec::scale(coefficient, r1.Vector)), v2)) return false;
    out.V1 = r1.Vector;
    out.V2 = v2;
    out.P = r1.Origin;
    return true;
}
static inline void basis(Vector viewUp, const OrthonormalPlane &plane, Ray &outX, Ray &outY) { // Plane.fs:82-97
    UnitVector up{0, 0, 0};
    Vec::unitise(viewUp, up);
    double v1Component = Vec::dot(plane.V1, up);
    double v2Component = Vec::dot(plane.V2, up);
    UnitVector v2{0, 0, 0}, v1{0, 0, 0};
    Vec::unitise(Vec::sum(Vec::scale(v1Component, plane.V1), Vec::scale(v2Component, plane.V2)), v2);
    Vec::unitise(Vec::sum(Vec::scale(v2Component, plane.V1), Vec::scale(-v1Component, plane.V2)), v1);
    outX = RayM::make(plane.P, v1);
    outY = RayM::make(plane.P, v2);
}
} // namespace PlaneM

// ------------------------------------------------------------------------------------------------
// LightRay.fs
// ------------------------------------------------------------------------------------------------
struct LightRay { Ray ray; Pixel Colour; }; // LightRay.fs:7-17

// ------------------------------------------------------------------------------------------------
// BoundingBox.fs
// ------------------------------------------------------------------------------------------------
struct BoundingBox { Point Min, Max; }; // BoundingBox.fs:3-8
struct Inv3 { double x, y, z; };

namespace BBox {
static inline double volume(const BoundingBox &box) { // BoundingBox.fs:13-16
    return (box.Max.x - box.Min.x) * (box.Max.y - box.Min.y) * (box.Max.z - box.Min.z);
}
static inline Inv3 inverseDirections(const Ray &ray) { // BoundingBox.fs:25-28
    return Inv3{1.0 / ray.Vector.x, 1.0 / ray.Vector.y, 1.0 / ray.Vector.z};
}
static inline bool hits(Inv3 inv, const Ray &ray, const BoundingBox &box) { // BoundingBox.fs:30-94
    double x = ray.Origin.x, y = ray.Origin.y, z = ray.Origin.z;
    double tMin = -std::numeric_limits<double>::infinity();
    double tMax = std::numeric_limits<double>::infinity();
    bool bailOut;
    {
        double t0 = (box.Min.x - x) * inv.x;
        double t1 = (box.Max.x - x) * inv.x;
        if (inv.x < 0.0) { double tmp = t1; t1 = t0; t0 = tmp; }
        tMin = (t0 > tMin) ? t0 : tMin;
        tMax = (t1 < tMax) ? t1 : tMax;
        bailOut = tMax < tMin || 0.0 >= tMax;
    }
    if (bailOut) return false;
    {
        double t0 = (box.Min.y - y) * inv.y;
        double t1 = (box.Max.y - y) * inv.y;
        if (inv.y < 0.0) { double tmp = t1; t1 = t0; t0 = tmp; }
        tMin = (t0 > tMin) ? t0 : tMin;
        tMax = (t1 < tMax) ? t1 : tMax;
        bailOut = tMax < tMin || 0.0 >= tMax;
    }
    if (bailOut) return false;
    double t0 = (box.Min.z - z) * inv.z;
    double t1 = (box.Max.z - z) * inv.z;
    if (inv.z < 0.0) { double tmp = t1; t1 = t0; t0 = tmp; }
    tMin = (t0 > tMin) ? t0 : tMin;
    tMax = (t1 < tMax) ? t1 : tMax;
    return tMax >= tMin && tMax >= 0.0;
}
// F# `min`/`max` on floats (BoundingBox.fs:96-108): `if e1 < e2 then e1 else e2` / `if e1 < e2 then e2 else e1`
// with NaN propagation in FSharp.Core; scene coordinates are never NaN.
static inline double fmin_(double a, double b) { return a < b ? a : b; }
static inline double fmax_(double a, double b) { return a < b ? b : a; }
static inline BoundingBox mergeTwo(const BoundingBox &i, const BoundingBox &j) { // BoundingBox.fs:96-108
    return BoundingBox{Pt::make(fmin_(i.Min.x, j.Min.x), fmin_(i.Min.y, j.Min.y), fmin_(i.Min.z, j.Min.z)),
                       Pt::make(fmax_(i.Max.x, j.Max.x), fmax_(i.Max.y, j.Max.y), fmax_(i.Max.z, j.Max.z))};
}
} // namespace BBox

// ------------------------------------------------------------------------------------------------
// Texture.fs (closures enumerated as rt_texture kinds; see include/rtfs_amd.h)
// ------------------------------------------------------------------------------------------------
struct TextureTable {
    std::vector<rt_texture> tex;
    std::vector<std::vector<uint8_t>> texels; // per texture (IMAGE only)
};

// Math.Acos / Math.Atan2 / Math.Sin (Sphere.fs:59-60, Texture.fs:58) are the platform's C runtime in .NET; libms agree with each
// other only to an ulp or so, and one ulp of (u, v) flips a truncating texel index.  The oracle takes the platform-independent
// value those approximate: the exact result rounded once, computed here in binary128 (libquadmath, ~113 bits) and rounded to
// double -- deliberately a different route from the HIP path's double-double Newton step (csrc/rt_trig.h).  glibc's results are
// within 1 ulp of these (tests/test_trig_cr.py, which also pins both routes to mpmath at 200 bits).
static inline double crAcos(double x) { return (double) acosq((__float128) x); }
static inline double crAtan2(double y, double x) { return (double) atan2q((__float128) y, (__float128) x); }
static inline double crSin(double x) { return (double) sinq((__float128) x); }

// Sphere.planeMapInverse (Sphere.fs:55-61)
static inline void planeMapInverse(double radius, Point centre, Point p, double &outPhi, double &outTheta) {
    Vector v = Vec::scale(1.0 / radius, Pt::differenceToThenFrom(p, centre));
    double theta = crAcos(-v.y);
    double phi = crAtan2(-v.z, v.x) + M_PI;
    outPhi = (phi / (2.0 * M_PI));
    outTheta = theta / M_PI;
}
// Sphere.planeMap (Sphere.fs:47-52)
static inline Point planeMap(double radius, Point centre, double phi, double theta) {
    theta = theta * M_PI;
    phi = phi * M_PI * 2.0 - M_PI;
    return Pt::sum(centre, Pt::make(radius * std::cos(phi) * std::sin(theta), -radius * std::cos(theta),
                                    -radius * std::sin(phi) * std::sin(theta)));
}

// `int (v * float (n - 1))` (Texture.fs:65-66) as an index into n texels: the reference's truncation for a coordinate in [0, 1].
// Outside it (Math.Acos is NaN for a point a rounding error off its sphere) the reference throws IndexOutOfRangeException; the
// defined behaviour, here and on the device: NaN and negative products take texel 0, products beyond the end the last texel.
static inline int texelIndex(double v, int n) {
    const double t = v * (double) (n - 1);
    if (!(t >= 0.0)) return 0;
    if (t >= (double) (n - 1)) return n - 1;
    return (int) t;
}
static inline uint8_t rampByte(double v) { // byte (v * 255.0); NaN -> 0 on both sides
    const double t = v * 255.0;
    return (t == t) ? (uint8_t) (int32_t) t : (uint8_t) 0;
}

// ParameterisedTexture.colourAt (Texture.fs:50-67); (x, y) = interpret p is evaluated by the caller once:
// every branch that needs it calls the same pure `interpret p`.
static Pixel paramColourAt(const TextureTable &tt, int id, double x, double y) {
    const rt_texture &t = tt.tex[(size_t) id];
    switch (t.kind) {
    case RT_TEXTURE_COLOUR: // Texture.fs:52
        return Pixel{t.rgb[0], t.rgb[1], t.rgb[2]};
    case RT_TEXTURE_UV_RAMP: { // Texture.fs:53-55 with the closures of RayTracing.App/SampleImages.fs:606-627
        uint8_t c[3];
        for (int k = 0; k < 3; ++k) {
            if (t.ramp_src[k] == RT_RAMP_U) c[k] = rampByte(x);      // byte (float x * 255.0)
            else if (t.ramp_src[k] == RT_RAMP_V) c[k] = rampByte(y); // byte (y * 255.0)
            else c[k] = t.rgb[k];
        }
        return Pixel{c[0], c[1], c[2]};
    }
    case RT_TEXTURE_CHECKERED: { // Texture.fs:56-62
        double sine = crSin(t.grid_size * x) * crSin(t.grid_size * y);
        if (Float::compare(sine, 0.0) == Comparison::Less) return paramColourAt(tt, t.even, x, y);
        return paramColourAt(tt, t.odd, x, y);
    }
    case RT_TEXTURE_IMAGE: { // Texture.fs:63-67
        int xi = texelIndex(1.0 - x, t.width);
        int yi = texelIndex(y, t.height);
        const uint8_t *px = &tt.texels[(size_t) id][((size_t) yi * (size_t) t.width + (size_t) xi) * 3];
        return Pixel{px[0], px[1], px[2]};
    }
    }
    return Colour::Black;
}

// Texture.colourAt (Texture.fs:12-15) for a style's texture: rgb when texture < 0, else
// ParameterisedTexture.toTexture (Sphere.planeMapInverse r c) tex (Texture.fs:69-72).
static Pixel textureColourAt(const TextureTable &tt, int texture, Pixel rgb, Point strike, double *uvOut = nullptr) {
    if (texture < 0) return rgb;
    const rt_texture &t = tt.tex[(size_t) texture];
    if (t.kind == RT_TEXTURE_COLOUR) return Pixel{t.rgb[0], t.rgb[1], t.rgb[2]}; // Texture.fs:71
    double u, v;
    planeMapInverse(t.map_radius, Pt::make(t.map_centre[0], t.map_centre[1], t.map_centre[2]), strike, u, v);
    if (uvOut) { uvOut[0] = u; uvOut[1] = v; }
    return paramColourAt(tt, texture, u, v);
}

// ------------------------------------------------------------------------------------------------
// Sphere.fs
// ------------------------------------------------------------------------------------------------
struct SphereStyle { // Sphere.fs:10-37 (FloatProducer members replaced by the path's stream)
    uint32_t kind;
    double albedo, fuzz, ior, prob;
    Pixel colour; // Texture.Colour / LightSourceCap pixel
    int texture;  // index or -1
};

struct Sphere { // Sphere.fs:302-310
    Point Centre;
    double Radius;
    double RadiusSquared;
    BoundingBox Box;
    SphereStyle Style;
};

namespace SphereM {
// Sphere.normal (Sphere.fs:65-66); ValueOption.get is asserted (radius >= 1e-4 in every scene).
static inline Ray normal(Point centre, Point p) {
    Ray r{p, Vector{0, 0, 0}};
    RayM::makePrime(p, Pt::differenceToThenFrom(p, centre), r);
    return r;
}

// Sphere.reflectWithoutFuzz (Sphere.fs:68-87)
static void reflectWithoutFuzz(const Ray &normal, Point strikePoint, LightRay &incomingLight) {
    OrthonormalPlane plane;
    if (!PlaneM::makeOrthonormalSpannedBy(normal, incomingLight.ray, plane)) {
        // Incoming ray is directly along the normal
        RayM::flipInPlace(incomingLight.ray);
        RayM::translateToIntersect(strikePoint, incomingLight.ray);
    } else {
        double normalComponent = -Vec::dot(plane.V1, incomingLight.ray.Vector);
        double tangentComponent = (Vec::dot(plane.V2, incomingLight.ray.Vector));
        Point dest = RayM::walkAlongRay(RayM::walkAlongRay(plane.P, plane.V1, normalComponent), plane.V2, tangentComponent);
        RayM::overwriteWithMake(strikePoint, Pt::differenceToThenFrom(dest, strikePoint), incomingLight.ray);
    }
}

// Sphere.addFuzz (Sphere.fs:89-104)
static void addFuzz(double fuzz, FloatProducer &rand, Point strikePoint, LightRay &reflected) {
    bool isDone = false;
    while (!isDone) {
        UnitVector offset = UV::random(rand);
        Point sphereCentre = RayM::walkAlong(reflected.ray, 1.0);
        Point target = RayM::walkAlongRay(sphereCentre, offset, fuzz);
        Vector newDirection = Pt::differenceToThenFrom(target, strikePoint);
        isDone = RayM::overwriteWithMake(strikePoint, newDirection, reflected.ray);
    }
}

// Sphere.refract (Sphere.fs:108-146)
static void refract(bool inside, const Ray &normal, Point strikePoint, double incomingCos, double index,
                    LightRay &incomingLight) {
    index = inside ? 1.0 / index : index / 1.0;
    OrthonormalPlane plane;
    if (!PlaneM::makeOrthonormalSpannedBy(normal, incomingLight.ray, plane)) {
        // Incoming ray was parallel to normal; pass straight through
        Vector vec = incomingLight.ray.Vector;
        RayM::overwriteWithMake(strikePoint, vec, incomingLight.ray);
        return;
    }
    double incomingSin = std::sqrt(1.0 - incomingCos * incomingCos);
    double outgoingSin = incomingSin / index;
    if (Float::compare(outgoingSin, 1.0) == Comparison::Greater) {
        reflectWithoutFuzz(normal, strikePoint, incomingLight);
        return;
    }
    double outgoingCos = std::sqrt(1.0 - outgoingSin * outgoingSin);
    Point outgoingPoint = RayM::walkAlong(RayM::make(RayM::walkAlong(normal, (-outgoingCos)), plane.V2), outgoingSin);
    Vector outgoingLine = Pt::differenceToThenFrom(outgoingPoint, strikePoint);
    RayM::overwriteWithMake(strikePoint, outgoingLine, incomingLight.ray);
}

// Math.Pow (x, 5.0) (Sphere.fs:290).  Math.Pow is the platform's C runtime pow: glibc on Linux, whose result is within
// 1 ulp but NOT always correctly rounded (measured: 0.08 % of x in [0,2] are 1 ulp off), the UCRT on Windows, etc.
// The oracle takes the platform-independent value those approximate: x^5 rounded once, computed here in binary128
// (5 multiplies, error < 2^-110) -- deliberately a different route from the HIP path's double-double product.
static inline double pow5(double x) {
    __float128 q = (__float128) x;
    return (double) (q * q * q * q * q);
}

// Sphere.reflection (Sphere.fs:150-300).  Returns true with `absorbed` set for ValueSome colour.
static bool reflection(const TextureTable &tt, const SphereStyle &style, Point centre, double radius, double radiusSquared,
                       bool flipped, LightRay &incomingLight, Point strikePoint, FloatProducer &rand, Pixel &absorbed) {
    bool inside = false;
    Ray normal = SphereM::normal(centre, strikePoint);
    switch (Float::compare(Vec::normSquared(Pt::differenceToThenFrom(centre, incomingLight.ray.Origin)), radiusSquared)) {
    case Comparison::Equal:
    case Comparison::Less:
        if (!flipped) { inside = true; RayM::flipInPlace(normal); }
        break;
    case Comparison::Greater:
        if (flipped) { inside = true; RayM::flipInPlace(normal); }
        break;
    }

    switch (style.kind) {
    case RT_SPHERE_LIGHT_SOURCE: { // Sphere.fs:185-189
        absorbed = PixelM::combine(incomingLight.Colour, textureColourAt(tt, style.texture, style.colour, strikePoint));
        return true;
    }
    case RT_SPHERE_LIGHT_SOURCE_CAP: { // Sphere.fs:190-200 (reads coordinate 0 despite the "Z" names)
        double circleCentreZCoord = Pt::coordinate(0, centre);
        double zCoordLowerBound = circleCentreZCoord + (radius - (radius / 4.0));
        double strikeZCoord = Pt::coordinate(0, strikePoint);
        if (Float::compare(strikeZCoord, zCoordLowerBound) == Comparison::Greater)
            absorbed = PixelM::combine(style.colour, incomingLight.Colour);
        else absorbed = Colour::Black;
        return true;
    }
    case RT_SPHERE_LAMBERT_REFLECTION: { // Sphere.fs:202-222
        Pixel newColour = PixelM::darken(style.albedo, PixelM::combine(incomingLight.Colour,
                                                       textureColourAt(tt, style.texture, style.colour, strikePoint)));
        incomingLight.Colour = newColour;
        Point sphereCentre = RayM::walkAlong(normal, 1.0);
        bool isDone = false;
        while (!isDone) {
            UnitVector offset = UV::random(rand);
            Point target = RayM::walkAlongRay(sphereCentre, offset, 1.0);
            Vector outputVec = Pt::differenceToThenFrom(target, strikePoint);
            isDone = RayM::overwriteWithMake(strikePoint, outputVec, incomingLight.ray);
        }
        return false;
    }
    case RT_SPHERE_PURE_REFLECTION: { // Sphere.fs:224-233
        Pixel darkened = PixelM::darken(style.albedo, PixelM::combine(incomingLight.Colour,
                                                      textureColourAt(tt, style.texture, style.colour, strikePoint)));
        reflectWithoutFuzz(normal, strikePoint, incomingLight);
        incomingLight.Colour = darkened;
        return false;
    }
    case RT_SPHERE_FUZZED_REFLECTION: { // Sphere.fs:235-246
        Pixel darkened = PixelM::darken(style.albedo, PixelM::combine(incomingLight.Colour,
                                                      textureColourAt(tt, style.texture, style.colour, strikePoint)));
        incomingLight.Colour = darkened;
        reflectWithoutFuzz(normal, strikePoint, incomingLight);
        addFuzz(style.fuzz, rand, strikePoint, incomingLight);
        return false;
    }
    case RT_SPHERE_DIELECTRIC: { // Sphere.fs:248-267
        Pixel newColour = PixelM::darken(style.albedo, PixelM::combine(incomingLight.Colour,
                                                       textureColourAt(tt, style.texture, style.colour, strikePoint)));
        double r = rand.Get();
        if (r > style.prob) {
            incomingLight.Colour = newColour;
            reflectWithoutFuzz(normal, strikePoint, incomingLight);
            return false;
        } else {
            double incomingCos = Vec::dot(incomingLight.ray.Vector, normal.Vector);
            refract(inside, normal, strikePoint, incomingCos, style.ior, incomingLight);
            incomingLight.Colour = newColour;
            return false;
        }
    }
    case RT_SPHERE_GLASS: { // Sphere.fs:269-300
        Pixel newColour = PixelM::darken(style.albedo, PixelM::combine(incomingLight.Colour,
                                                       textureColourAt(tt, style.texture, style.colour, strikePoint)));
        double incomingCos = Vec::dot(UV::flip(incomingLight.ray.Vector), normal.Vector);
        double r = rand.Get();
        double reflectionProb;
        {
            double sphereRefractance = inside ? 1.0 / style.ior : style.ior;
            double param = (1.0 - sphereRefractance) / (1.0 + sphereRefractance);
            param = param * param;
            reflectionProb = param + (1.0 - param) * (pow5((1.0 - incomingCos))); // `** 5.0` = Math.Pow, see pow5 above
        }
        if (r < reflectionProb) {
            reflectWithoutFuzz(normal, strikePoint, incomingLight);
            incomingLight.Colour = newColour;
            return false;
        } else {
            refract(inside, normal, strikePoint, incomingCos, style.ior, incomingLight);
            incomingLight.Colour = newColour;
            return false;
        }
    }
    }
    absorbed = Colour::Black;
    return true;
}

// Sphere.make (Sphere.fs:325-337)
static Sphere make(const SphereStyle &style, Point centre, double radius) {
    double radiusSquared = radius * radius;
    Sphere s;
    s.Style = style;
    s.Centre = centre;
    s.Radius = radius;
    s.RadiusSquared = radiusSquared;
    s.Box = BoundingBox{Pt::sum(centre, Pt::make(-radius, -radius, -radius)), Pt::sum(centre, Pt::make(radius, radius, radius))};
    return s;
}

// Sphere.liesOn (Sphere.fs:341-343)
static inline bool liesOn(Point point, const Sphere &sphere) {
    return Float::equal(Vec::normSquared(Pt::differenceToThenFrom(point, sphere.Centre)), sphere.RadiusSquared);
}

// Sphere.firstIntersection (Sphere.fs:349-386)
static inline bool firstIntersection(const Sphere &sphere, const Ray &ray, double &out) {
    Vector difference = Pt::differenceToThenFrom(ray.Origin, sphere.Centre);
    double b = (Vec::dot(ray.Vector, difference));
    double c = (Vec::normSquared(difference)) - sphere.RadiusSquared;
    double discriminantOverFour = (b * b - c);
    bool some = false;
    double i = 0.0;
    switch (Float::compare(discriminantOverFour, 0.0)) {
    case Comparison::Equal: some = true; i = (-b); break;
    case Comparison::Less: some = false; break;
    case Comparison::Greater: {
        double intermediate = std::sqrt(discriminantOverFour);
        double i1 = intermediate - b;
        double i2 = -(b + intermediate);
        bool i1Pos = Float::positive(i1);
        bool i2Pos = Float::positive(i2);
        if (i1Pos && i2Pos) {
            switch (Float::compare(i1, i2)) {
            case Comparison::Less: i = i1; break;
            case Comparison::Greater: i = i2; break;
            case Comparison::Equal: i = i1; break;
            }
            some = true;
        } else if (i1Pos) { some = true; i = i1; }
        else if (i2Pos) { some = true; i = i2; }
        else some = false;
        break;
    }
    }
    if (!some) return false;
    if (Float::positive(i)) { out = i; return true; }
    return false;
}
} // namespace SphereM

// ------------------------------------------------------------------------------------------------
// InfinitePlane.fs
// ------------------------------------------------------------------------------------------------
struct InfinitePlaneStyle { uint32_t kind; double albedo, fuzz; Pixel colour; int texture; }; // InfinitePlane.fs:3-13
struct InfinitePlane { InfinitePlaneStyle Style; UnitVector Normal; Point P; };               // InfinitePlane.fs:101-106

namespace PlaneObj {
// InfinitePlane.pureOutgoing (InfinitePlane.fs:18-38)
static Ray pureOutgoing(Point strikePoint, UnitVector normal, const Ray &incomingRay) {
    OrthonormalPlane plane;
    if (!PlaneM::makeOrthonormalSpannedBy(RayM::make(strikePoint, normal), incomingRay, plane)) {
        return RayM::parallelTo(strikePoint, RayM::flip(incomingRay));
    }
    double normalComponent = -(Vec::dot(plane.V1, incomingRay.Vector));
    double tangentComponent = (Vec::dot(plane.V2, incomingRay.Vector));
    Point s = RayM::walkAlong(RayM::make(RayM::walkAlong(RayM::make(plane.P, plane.V1), normalComponent), plane.V2), tangentComponent);
    Ray out{strikePoint, incomingRay.Vector};
    RayM::makePrime(strikePoint, Pt::differenceToThenFrom(s, strikePoint), out); // ValueOption.get asserted
    return out;
}
// InfinitePlane.newColour (InfinitePlane.fs:40-41)
static inline Pixel newColour(Pixel incomingColour, double albedo, Pixel colour) {
    return PixelM::darken(albedo, PixelM::combine(incomingColour, colour));
}
// InfinitePlane.reflection (InfinitePlane.fs:43-99).  `failed` reports the ValueOption.get at :86 throwing
// (the reference would crash; oracle and HIP path both end the path with Black -- DESIGN.md "Undefined cases").
static bool reflection(const TextureTable &tt, const InfinitePlaneStyle &style, Point pointOnPlane, UnitVector normal,
                       LightRay &incomingRay, Point strikePoint, FloatProducer &rand, Pixel &absorbed) {
    (void) pointOnPlane;
    switch (style.kind) {
    case RT_PLANE_LIGHT_SOURCE: // InfinitePlane.fs:52-56
        absorbed = PixelM::combine(incomingRay.Colour, textureColourAt(tt, style.texture, style.colour, strikePoint));
        return true;
    case RT_PLANE_FUZZED_REFLECTION: { // InfinitePlane.fs:58-76
        Pixel nc = newColour(incomingRay.Colour, style.albedo, style.colour);
        Ray pure = pureOutgoing(strikePoint, normal, incomingRay.ray);
        Ray outgoing{};
        bool have = false;
        while (!have) {
            UnitVector offset = UV::random(rand);
            Point sphereCentre = RayM::walkAlong(pure, 1.0);
            Point target = RayM::walkAlong(RayM::make(sphereCentre, offset), style.fuzz);
            have = RayM::makePrime(strikePoint, Pt::differenceToThenFrom(target, strikePoint), outgoing);
        }
        incomingRay.Colour = nc;
        incomingRay.ray = outgoing;
        return false;
    }
    case RT_PLANE_LAMBERT_REFLECTION: { // InfinitePlane.fs:78-93
        Point sphereCentre = RayM::walkAlong(RayM::make(strikePoint, normal), 1.0);
        UnitVector offset = UV::random(rand);
        Point target = RayM::walkAlong(RayM::make(sphereCentre, offset), 1.0);
        Ray outgoing{};
        if (!RayM::makePrime(strikePoint, Pt::differenceToThenFrom(target, strikePoint), outgoing)) {
            absorbed = Colour::Black; // the reference throws here (ValueOption.get, :86)
            return true;
        }
        Pixel nc = PixelM::darken(style.albedo, PixelM::combine(incomingRay.Colour, style.colour));
        incomingRay.Colour = nc;
        incomingRay.ray = outgoing;
        return false;
    }
    case RT_PLANE_PURE_REFLECTION: // InfinitePlane.fs:95-99
        incomingRay.Colour = newColour(incomingRay.Colour, style.albedo, style.colour);
        incomingRay.ray = pureOutgoing(strikePoint, normal, incomingRay.ray);
        return false;
    }
    absorbed = Colour::Black;
    return true;
}
// InfinitePlane.intersection (InfinitePlane.fs:125-136)
static inline bool intersection(const InfinitePlane &plane, const Ray &ray, double &out) {
    UnitVector rayVec = ray.Vector;
    double denominator = Vec::dot(plane.Normal, rayVec);
    if (Float::equal(denominator, 0.0)) return false;
    double t = (Vec::dot(plane.Normal, Pt::differenceToThenFrom(plane.P, ray.Origin))) / denominator;
    if (Float::positive(t)) { out = t; return true; }
    return false;
}
} // namespace PlaneObj

// ------------------------------------------------------------------------------------------------
// Hittable.fs
// ------------------------------------------------------------------------------------------------
struct Hittable { // Hittable.fs:3-6
    uint32_t kind;
    Sphere sphere;
    InfinitePlane plane;
    int index; // position in the array handed to Scene.make
};

struct Counters { uint64_t rays = 0, aabb = 0, prim = 0, refl = 0, samples = 0; };

namespace HittableM {
static inline bool hits(const Ray &ray, const Hittable &h, double &out, Counters &cnt) { // Hittable.fs:27-31
    cnt.prim++;
    if (h.kind == RT_HITTABLE_INFINITE_PLANE) return PlaneObj::intersection(h.plane, ray, out);
    return SphereM::firstIntersection(h.sphere, ray, out);
}
static inline bool reflection(const TextureTable &tt, const Hittable &h, LightRay &incoming, Point strikePoint,
                              FloatProducer &rand, Pixel &absorbed, Counters &cnt) { // Hittable.fs:8-12
    cnt.refl++;
    if (h.kind == RT_HITTABLE_INFINITE_PLANE)
        return PlaneObj::reflection(tt, h.plane.Style, h.plane.P, h.plane.Normal, incoming, strikePoint, rand, absorbed);
    const Sphere &s = h.sphere;
    // Sphere.Reflection member (Sphere.fs:315-323): flipped = Float.compare this.Radius 0.0 = Less
    return SphereM::reflection(tt, s.Style, s.Centre, s.Radius, s.RadiusSquared,
                               Float::compare(s.Radius, 0.0) == Comparison::Less, incoming, strikePoint, rand, absorbed);
}
} // namespace HittableM

// ------------------------------------------------------------------------------------------------
// BoundingBoxTree.fs
// ------------------------------------------------------------------------------------------------
struct BoundingBoxTree { // BoundingBoxTree.fs:3-5
    bool leaf;
    const Hittable *hittable; // Leaf
    BoundingBox box;          // Leaf box | Branch `all`
    std::unique_ptr<BoundingBoxTree> left, right;
};

namespace BBTree {
typedef std::pair<const Hittable *, BoundingBox> Item;

static BoundingBox mergeAll(const std::vector<Item> &boxes) { // BoundingBox.merge (BoundingBox.fs:110-114): Array.reduce mergeTwo
    BoundingBox acc = boxes[0].second;
    for (size_t i = 1; i < boxes.size(); ++i) acc = BBox::mergeTwo(acc, boxes[i].second);
    return acc;
}

static std::unique_ptr<BoundingBoxTree> go(const std::vector<Item> &boxes) { // BoundingBoxTree.fs:14-41
    BoundingBox boundAll = mergeAll(boxes);
    auto node = std::make_unique<BoundingBoxTree>();
    if (boxes.size() == 1) {
        node->leaf = true;
        node->hittable = boxes[0].first;
        node->box = boxes[0].second;
        return node;
    }
    auto mkLeaf = [](const Item &it) {
        auto l = std::make_unique<BoundingBoxTree>();
        l->leaf = true;
        l->hittable = it.first;
        l->box = it.second;
        return l;
    };
    node->leaf = false;
    node->hittable = nullptr;
    node->box = boundAll;
    if (boxes.size() == 2) {
        node->left = mkLeaf(boxes[0]);
        node->right = mkLeaf(boxes[1]);
        return node;
    }
    std::vector<Item> bestLeft, bestRight;
    double bestCost = 0.0;
    for (int axis = 0; axis < 3; ++axis) {
        // Array.sortBy is an unstable introsort in .NET; ties (e.g. every small sphere has Min.y = 0.0) have no
        // defined order there.  DEFINED here as a stable sort (DESIGN.md "Tree shape"); the closest hit does not
        // depend on the tree shape, only the visit counters do.
        std::vector<Item> sorted = boxes;
        std::stable_sort(sorted.begin(), sorted.end(), [axis](const Item &a, const Item &b) {
            return Pt::coordinate(axis, a.second.Min) < Pt::coordinate(axis, b.second.Min);
        });
        size_t half = sorted.size() / 2;
        std::vector<Item> leftHalf(sorted.begin(), sorted.begin() + (long) half + 1); // boxes.[0 .. n/2]
        std::vector<Item> rightHalf(sorted.begin() + (long) half + 1, sorted.end());  // boxes.[n/2+1 ..]
        double cost = BBox::volume(mergeAll(leftHalf)) + BBox::volume(mergeAll(rightHalf));
        if (axis == 0 || cost < bestCost) { // Array.minBy keeps the first minimum
            bestCost = cost;
            bestLeft.swap(leftHalf);
            bestRight.swap(rightHalf);
        }
    }
    node->left = go(bestLeft);
    node->right = go(bestRight);
    return node;
}
} // namespace BBTree

// ------------------------------------------------------------------------------------------------
// Scene.fs
// ------------------------------------------------------------------------------------------------
struct Scene { // Scene.fs:5-10
    std::vector<Hittable> objects;          // owns every Hittable, original order
    std::vector<const Hittable *> unbounded; // UnboundedObjects
    std::unique_ptr<BoundingBoxTree> tree;   // BoundingBoxes
    TextureTable textures;
};

struct Best { double bestFloat; const Hittable *bestObject; double bestLength; };

namespace SceneM {
// Scene.bestCandidate (Scene.fs:30-60)
static Best bestCandidate(Inv3 inv, const Ray &ray, Best best, const BoundingBoxTree &box, Counters &cnt) {
    if (box.leaf) {
        cnt.aabb++;
        if (BBox::hits(inv, ray, box.box)) {
            double point;
            if (!HittableM::hits(ray, *box.hittable, point, cnt)) return best;
            double a = point * point;
            if (a < best.bestFloat) return Best{a, box.hittable, point};
            return best;
        }
        return best;
    }
    cnt.aabb++;
    if (BBox::hits(inv, ray, box.box)) {
        Best b = bestCandidate(inv, ray, best, *box.left, cnt);
        return bestCandidate(inv, ray, b, *box.right, cnt);
    }
    return best;
}

// Independent cross-check of the tree walk (SURVEY.md 8c pin iv), NOT the reference's algorithm: test every bounded sphere,
// no boxes.  Equal to the tree walk except for rays that graze a sphere within the 1e-8 discriminant band outside its box.
static bool g_bruteForce = false;
static void collectLeaves(const BoundingBoxTree &t, std::vector<const Hittable *> &out) {
    if (t.leaf) { out.push_back(t.hittable); return; }
    collectLeaves(*t.left, out);
    collectLeaves(*t.right, out);
}

// Scene.hitObject (Scene.fs:62-91)
static bool hitObject(const Scene &s, const Ray &ray, const Hittable *&outObj, Point &outStrike, Counters &cnt) {
    cnt.rays++;
    Best best{std::numeric_limits<double>::infinity(), nullptr, std::numeric_limits<double>::quiet_NaN()};
    if (s.tree && g_bruteForce) {
        std::vector<const Hittable *> leaves;
        collectLeaves(*s.tree, leaves); // same visiting order as the walk
        for (const Hittable *h : leaves) {
            double point;
            if (HittableM::hits(ray, *h, point, cnt)) {
                double a = point * point;
                if (a < best.bestFloat) best = Best{a, h, point};
            }
        }
    } else
    if (s.tree) best = bestCandidate(BBox::inverseDirections(ray), ray, best, *s.tree, cnt);
    for (const Hittable *i : s.unbounded) {
        double point;
        if (HittableM::hits(ray, *i, point, cnt)) {
            double a = point * point;
            if (Float::compare(a, best.bestFloat) == Comparison::Less) {
                best.bestFloat = a;
                best.bestObject = i;
                best.bestLength = point;
            }
        }
    }
    if (std::isnan(best.bestLength)) return false;
    outObj = best.bestObject;
    outStrike = RayM::walkAlong(ray, best.bestLength);
    return true;
}

// Scene.traceRay (Scene.fs:93-114)
static Pixel traceRay(int maxCount, const Scene &scene, LightRay &ray, FloatProducer &rand, Counters &cnt) {
    int bounces = 0;
    Pixel result = Colour::Black;
    bool isDone = false;
    while (bounces <= maxCount && !isDone) {
        const Hittable *object;
        Point strikePoint;
        if (!hitObject(scene, ray.ray, object, strikePoint, cnt)) {
            isDone = true;
        } else {
            Pixel colour;
            if (HittableM::reflection(scene.textures, *object, ray, strikePoint, rand, colour, cnt)) {
                isDone = true;
                result = colour;
            } else bounces = bounces + 1;
        }
    }
    return !isDone ? Colour::HotPink : result;
}

struct Camera { // Camera.fs:3-28
    double ViewportHeight, ViewportWidth;
    Ray View, ViewportXAxis, ViewportYAxis;
    double FocalLength;
    int SamplesPerPixel, BounceDepth;
};

// Scene.traceOnce (Scene.fs:118-155); `rand` is the (pixel, sample) stream, shared with the materials.
static void traceOnce(const Scene &scene, FloatProducer &rand, const Camera &camera, int maxWidthCoord, int maxHeightCoord,
                      int row, int col, PixelStats &stats, Counters &cnt) {
    cnt.samples++;
    double rand1, rand2;
    rand.GetTwo(rand1, rand2);
    double landingPoint = (((double) col + rand1) * camera.ViewportWidth) / (double) maxWidthCoord;
    Point pointOnXAxis = RayM::walkAlong(camera.ViewportXAxis, landingPoint);
    double walkDistance = (((double) row + rand2) * camera.ViewportHeight) / (double) maxHeightCoord;
    Point endPoint = RayM::walkAlongRay(pointOnXAxis, camera.ViewportYAxis.Vector, walkDistance);
    Ray ray{};
    if (!RayM::makePrime(camera.View.Origin, Pt::differenceToThenFrom(endPoint, camera.View.Origin), ray)) {
        PixelStatsM::add(Colour::Black, stats); // the reference throws (ValueOption.get, Scene.fs:144)
        return;
    }
    LightRay initialRay{ray, Colour::White};
    Pixel result = traceRay(camera.BounceDepth, scene, initialRay, rand, cnt);
    PixelStatsM::add(result, stats);
}

// Scene.renderPixel (Scene.fs:157-194).  Sample i of the pixel uses stream (seed, pixel, i).
static Pixel renderPixel(const Scene &scene, uint64_t seed, uint64_t pixelIndex, const Camera &camera, int maxWidthCoord,
                         int maxHeightCoord, int row, int col, PixelStats &stats, bool &early, Counters &cnt) {
    stats = PixelStatsM::empty();
    uint32_t sample = 0;
    auto once = [&]() {
        FloatProducer rand = streamFor(seed, pixelIndex, sample++);
        traceOnce(scene, rand, camera, maxWidthCoord, maxHeightCoord, row, col, stats, cnt);
    };
    int firstTrial = std::min(5, (camera.SamplesPerPixel / 2));
    for (int i = 0; i <= firstTrial; ++i) once();
    Pixel oldMean = PixelStatsM::mean(stats);
    for (int i = 1; i <= firstTrial; ++i) once();
    Pixel newMean = PixelStatsM::mean(stats);
    int difference = PixelM::difference(newMean, oldMean);
    if (difference == 0) {
        early = true;
        return newMean;
    }
    early = false;
    for (int i = 1; i <= (camera.SamplesPerPixel - 2 * firstTrial - 1); ++i) once();
    return PixelStatsM::mean(stats);
}
} // namespace SceneM

// Camera.makeBasic (Camera.fs:34-59)
static SceneM::Camera cameraMakeBasic(int samplesPerPixel, double focalLength, double aspectRatio, Point origin,
                                      UnitVector viewDirection, Vector viewUp) {
    double height = 2.0;
    Ray view = RayM::make(origin, viewDirection);
    Point corner = RayM::walkAlong(view, focalLength);
    OrthonormalPlane viewPlane = PlaneM::makeNormalTo(corner, viewDirection);
    Ray xAxis, yAxis;
    PlaneM::basis(viewUp, viewPlane, xAxis, yAxis);
    SceneM::Camera c;
    c.FocalLength = focalLength;
    c.ViewportHeight = height;
    c.ViewportWidth = aspectRatio * height;
    c.View = view;
    c.ViewportXAxis = xAxis;
    c.ViewportYAxis = yAxis;
    c.SamplesPerPixel = samplesPerPixel;
    c.BounceDepth = 150;
    return c;
}

// PixelOutput.correct (ImageOutput.fs:11-18)
static inline uint8_t gammaCorrect(uint8_t b) {
    int i = (int) std::rint(std::sqrt((double) b / 255.0) * 255.0);
    if (i == 256) i = 255;
    return (uint8_t) i;
}

// ImageOutput.writePpm (ImageOutput.fs:163-197) into a string
static std::string formatPpm(const uint8_t *rgb, int rows, int cols, bool gamma) {
    std::string s;
    s += "P3\n";
    s += std::to_string(cols) + " " + std::to_string(rows) + "\n";
    s += "255\n";
    auto px = [&](int r, int c) {
        const uint8_t *p = rgb + ((size_t) r * (size_t) cols + (size_t) c) * 3;
        uint8_t R = p[0], G = p[1], B = p[2];
        if (gamma) { R = gammaCorrect(R); G = gammaCorrect(G); B = gammaCorrect(B); }
        s += std::to_string((int) R) + " " + std::to_string((int) G) + " " + std::to_string((int) B);
    };
    for (int r = 0; r < rows; ++r) {
        for (int c = 0; c < cols - 1; ++c) { px(r, c); s += " "; }
        px(r, cols - 1);
        if (r != rows - 1) s += "\n";
    }
    return s;
}

// ImageOutput.writeAsciiInt (ImageOutput.fs:115-129): emits nothing for 0
static void writeAsciiInt(std::string &writer, int i) {
    int places = 0, tmp = i, pw = 1;
    while (tmp > 0) { tmp = tmp / 10; pw = pw * 10; places = places + 1; }
    (void) places;
    pw = pw / 10;
    while (pw > 0) { writer.push_back((char) ((uint8_t) ((i / pw) % 10) + 48)); pw = pw / 10; }
}
// ImageOutput.resume's file body (ImageOutput.fs:144-158)
static std::string formatPixelMap(const uint8_t *rgb, int rows, int cols) {
    std::string s;
    for (int rowNum = 0; rowNum < rows; ++rowNum)
        for (int colNum = 0; colNum < cols; ++colNum) {
            const uint8_t *p = rgb + ((size_t) rowNum * (size_t) cols + (size_t) colNum) * 3;
            writeAsciiInt(s, rowNum);
            s.push_back((char) 44);
            writeAsciiInt(s, colNum);
            s.push_back((char) 10);
            s.push_back((char) p[0]); s.push_back((char) p[1]); s.push_back((char) p[2]);
        }
    return s;
}
// ImageOutput.consumeAsciiInteger (ImageOutput.fs:46-66) + readPixelMap's `go` (ImageOutput.fs:68-106)
static bool consumeAsciiInteger(const uint8_t *d, size_t n, size_t &pos, int &out) {
    int answer = 0;
    for (;;) {
        int i = pos < n ? (int) d[pos++] : -1;
        if (i < 0) return false;
        if (48 <= i && i <= 57) answer = (10 * answer + (i - 48));
        else { out = answer; return true; }
    }
}
static long parsePixelMap(const uint8_t *d, size_t n, int rows, int cols, uint8_t *rgb, uint8_t *present) {
    size_t pos = 0;
    long count = 0;
    for (;;) {
        int row, col;
        if (!consumeAsciiInteger(d, n, pos, row)) return count;
        if (!consumeAsciiInteger(d, n, pos, col)) return count;
        int r = pos < n ? (int) d[pos++] : -1;
        if (r < 0) return count;
        int g = pos < n ? (int) d[pos++] : -1;
        if (g == -1) return count;
        int b = pos < n ? (int) d[pos++] : -1;
        if (b == -1) return count;
        if (row >= rows || col >= cols) return -1; // dict.[row].[col] would throw
        size_t o = (size_t) row * (size_t) cols + (size_t) col;
        rgb[o * 3] = (uint8_t) r; rgb[o * 3 + 1] = (uint8_t) g; rgb[o * 3 + 2] = (uint8_t) b;
        if (present) present[o] = 1;
        ++count;
    }
}

// ---- conversions from the ABI structs ----------------------------------------------------------
static Point P3(const double *p) { return Point{p[0], p[1], p[2]}; }
static Vector V3(const double *p) { return Vector{p[0], p[1], p[2]}; }

static SceneM::Camera fromAbi(const rt_camera &c) {
    SceneM::Camera o;
    o.ViewportHeight = c.viewport_height;
    o.ViewportWidth = c.viewport_width;
    o.View = Ray{P3(c.view_origin), V3(c.view_dir)};
    o.ViewportXAxis = Ray{P3(c.xaxis_origin), V3(c.xaxis_dir)};
    o.ViewportYAxis = Ray{P3(c.yaxis_origin), V3(c.yaxis_dir)};
    o.FocalLength = c.focal_length;
    o.SamplesPerPixel = c.samples_per_pixel;
    o.BounceDepth = c.bounce_depth;
    return o;
}
static void toAbi(const SceneM::Camera &c, rt_camera *o) {
    auto st = [](double *d, double x, double y, double z) { d[0] = x; d[1] = y; d[2] = z; };
    st(o->view_origin, c.View.Origin.x, c.View.Origin.y, c.View.Origin.z);
    st(o->view_dir, c.View.Vector.x, c.View.Vector.y, c.View.Vector.z);
    st(o->xaxis_origin, c.ViewportXAxis.Origin.x, c.ViewportXAxis.Origin.y, c.ViewportXAxis.Origin.z);
    st(o->xaxis_dir, c.ViewportXAxis.Vector.x, c.ViewportXAxis.Vector.y, c.ViewportXAxis.Vector.z);
    st(o->yaxis_origin, c.ViewportYAxis.Origin.x, c.ViewportYAxis.Origin.y, c.ViewportYAxis.Origin.z);
    st(o->yaxis_dir, c.ViewportYAxis.Vector.x, c.ViewportYAxis.Vector.y, c.ViewportYAxis.Vector.z);
    o->viewport_width = c.ViewportWidth;
    o->viewport_height = c.ViewportHeight;
    o->focal_length = c.FocalLength;
    o->samples_per_pixel = c.SamplesPerPixel;
    o->bounce_depth = c.BounceDepth;
}

} // namespace orc

// =================================================================================================
// C entry points (loaded with ctypes by tests/ and bench.py's cpu_baseline leg only)
// =================================================================================================
using namespace orc;

struct orc_scene { Scene s; int depth = 0; int nodes = 0; };

static thread_local std::string g_err;

extern "C" {

const char *orc_last_error(void) { return g_err.c_str(); }

int orc_camera_make_basic(int32_t spp, double focal, double aspect, const double origin[3], const double view_dir[3],
                          const double view_up[3], rt_camera *out) {
    SceneM::Camera c = cameraMakeBasic(spp, focal, aspect, P3(origin), V3(view_dir), V3(view_up));
    toAbi(c, out);
    return RT_OK;
}

// Scene.make (Scene.fs:15-28)
int orc_scene_create(const rt_hittable *h, size_t n, const rt_texture *tex, size_t ntex, orc_scene **out) {
    auto sc = std::make_unique<orc_scene>();
    Scene &s = sc->s;
    s.textures.tex.assign(tex, tex + ntex);
    s.textures.texels.resize(ntex);
    for (size_t i = 0; i < ntex; ++i) {
        if (tex[i].kind == RT_TEXTURE_IMAGE) {
            if (!tex[i].texels || tex[i].width <= 0 || tex[i].height <= 0) { g_err = "image texture without texels"; return RT_ERR_INVALID_ARGUMENT; }
            size_t bytes = (size_t) tex[i].width * (size_t) tex[i].height * 3;
            s.textures.texels[i].assign(tex[i].texels, tex[i].texels + bytes);
        }
    }
    s.objects.resize(n);
    for (size_t i = 0; i < n; ++i) {
        Hittable &o = s.objects[i];
        o.kind = h[i].kind;
        o.index = (int) i;
        Pixel rgb{h[i].rgb[0], h[i].rgb[1], h[i].rgb[2]};
        if (h[i].kind == RT_HITTABLE_INFINITE_PLANE) {
            o.plane.Style = InfinitePlaneStyle{h[i].style, h[i].albedo, h[i].fuzz, rgb, h[i].texture};
            o.plane.Normal = V3(h[i].normal);
            o.plane.P = P3(h[i].point);
            o.sphere = Sphere{};
        } else {
            SphereStyle st{h[i].style, h[i].albedo, h[i].fuzz, h[i].ior, h[i].prob, rgb, h[i].texture};
            o.sphere = SphereM::make(st, P3(h[i].point), h[i].radius);
            o.plane = InfinitePlane{};
        }
    }
    std::vector<BBTree::Item> bounded;
    for (size_t i = 0; i < n; ++i) { // Array.partition keeps the order on both sides
        if (s.objects[i].kind == RT_HITTABLE_SPHERE) bounded.emplace_back(&s.objects[i], s.objects[i].sphere.Box);
        else s.unbounded.push_back(&s.objects[i]);
    }
    if (!bounded.empty()) s.tree = BBTree::go(bounded);
    *out = sc.release();
    return RT_OK;
}

void orc_scene_destroy(orc_scene *s) { delete s; }

// Flatten in DFS pre-order for comparison with rt_scene_get_tree.
static void flattenRec(const BoundingBoxTree &t, std::vector<int32_t> &skip, std::vector<int32_t> &prim,
                       std::vector<double> &boxes, int depth, int &maxDepth) {
    size_t me = skip.size();
    skip.push_back(0);
    prim.push_back(t.leaf ? t.hittable->index : -1);
    const BoundingBox &b = t.box;
    double v[6] = {b.Min.x, b.Max.x, b.Min.y, b.Max.y, b.Min.z, b.Max.z};
    boxes.insert(boxes.end(), v, v + 6);
    if (depth > maxDepth) maxDepth = depth;
    if (!t.leaf) {
        flattenRec(*t.left, skip, prim, boxes, depth + 1, maxDepth);
        flattenRec(*t.right, skip, prim, boxes, depth + 1, maxDepth);
    }
    skip[me] = (int32_t) skip.size();
}

int orc_scene_get_tree(const orc_scene *sc, int32_t *n_nodes, int32_t *depth, int32_t *skip, int32_t *prim, double *boxes) {
    std::vector<int32_t> sk, pr;
    std::vector<double> bx;
    int md = 0;
    if (sc->s.tree) flattenRec(*sc->s.tree, sk, pr, bx, 1, md);
    if (n_nodes) *n_nodes = (int32_t) sk.size();
    if (depth) *depth = md;
    if (skip) std::memcpy(skip, sk.data(), sk.size() * sizeof(int32_t));
    if (prim) std::memcpy(prim, pr.data(), pr.size() * sizeof(int32_t));
    if (boxes) std::memcpy(boxes, bx.data(), bx.size() * sizeof(double));
    return RT_OK;
}

// Scene.render (Scene.fs:196-236) for image rows row_first + i*row_stride, i < n_rows, on n_threads host threads.
int orc_render(const orc_scene *sc, const rt_camera *cam, int32_t max_w, int32_t max_h, uint64_t seed, int32_t row_first,
               int32_t row_stride, int32_t n_rows, int32_t n_threads, int32_t *accum, uint8_t *rgb, rt_stats *stats) {
    if (max_w <= 0 || max_h <= 0 || row_stride <= 0 || n_rows < 0) { g_err = "bad geometry"; return RT_ERR_INVALID_ARGUMENT; }
    const Scene &s = sc->s;
    SceneM::Camera camera = fromAbi(*cam);
    int rowsIter = 2 * max_h + 1; // Scene.fs:208
    int colsIter = 2 * max_w + 1; // Scene.fs:209
    if (row_first < 0 || (n_rows > 0 && row_first + (n_rows - 1) * row_stride >= rowsIter)) { g_err = "rows out of range"; return RT_ERR_INVALID_ARGUMENT; }
    if (n_threads <= 0) n_threads = 1;
    auto t0 = std::chrono::steady_clock::now();
    // Work units are blocks of 64 consecutive pixels of the shard (not whole rows), so that every host core has work
    // even when the shard has few rows; (pixel, sample) streams make the result independent of the schedule.
    const int64_t nPixels = (int64_t) n_rows * (int64_t) colsIter;
    const int64_t unitPixels = 64;
    const int64_t nUnits = (nPixels + unitPixels - 1) / unitPixels;
    std::atomic<int64_t> next{0};
    std::vector<Counters> cnts((size_t) n_threads);
    std::vector<uint64_t> earlies((size_t) n_threads, 0);
    auto worker = [&](int tid) {
        Counters cnt; // thread-local while rendering (adjacent per-thread slots would false-share their cache lines)
        uint64_t earlyLocal = 0;
        for (;;) {
            int64_t u = next.fetch_add(1);
            if (u >= nUnits) break;
            int64_t lpEnd = std::min(nPixels, (u + 1) * unitPixels);
            for (int64_t lp = u * unitPixels; lp < lpEnd; ++lp) {
                int i = (int) (lp / colsIter);
                int c = (int) (lp - (int64_t) i * colsIter);
                int r = row_first + i * row_stride;
                int row = max_h - r - 1; // Scene.fs:219
                int col = c - max_w;     // Scene.fs:226
                PixelStats st;
                bool early = false;
                uint64_t pixelIndex = (uint64_t) r * (uint64_t) colsIter + (uint64_t) c;
                Pixel p = SceneM::renderPixel(s, seed, pixelIndex, camera, max_w, max_h, row, col, st, early, cnt);
                size_t o = (size_t) lp;
                if (accum) { accum[o * 4 + 0] = st.Count; accum[o * 4 + 1] = st.SumRed; accum[o * 4 + 2] = st.SumGreen; accum[o * 4 + 3] = st.SumBlue; }
                if (rgb) { rgb[o * 3 + 0] = p.Red; rgb[o * 3 + 1] = p.Green; rgb[o * 3 + 2] = p.Blue; }
                if (early) earlyLocal++;
            }
        }
        cnts[(size_t) tid] = cnt;
        earlies[(size_t) tid] = earlyLocal;
    };
    if (n_threads == 1) worker(0);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < n_threads; ++t) th.emplace_back(worker, t);
        for (auto &t : th) t.join();
    }
    auto t1 = std::chrono::steady_clock::now();
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        for (int t = 0; t < n_threads; ++t) {
            stats->rays += cnts[(size_t) t].rays;
            stats->aabb_tests += cnts[(size_t) t].aabb;
            stats->prim_tests += cnts[(size_t) t].prim;
            stats->reflections += cnts[(size_t) t].refl;
            stats->samples += cnts[(size_t) t].samples;
            stats->pixels_early += earlies[(size_t) t];
        }
        stats->pixels = (uint64_t) n_rows * (uint64_t) colsIter;
        stats->total_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
        stats->kernel_ms = stats->total_ms;
    }
    return RT_OK;
}

void orc_set_brute_force(int on) { SceneM::g_bruteForce = on != 0; }

uint8_t orc_gamma_correct(uint8_t b) { return gammaCorrect(b); }

int64_t orc_format_ppm(const uint8_t *rgb, int32_t rows, int32_t cols, int32_t gamma, char *out, size_t cap) {
    std::string s = formatPpm(rgb, rows, cols, gamma != 0);
    if (out && cap > s.size()) { std::memcpy(out, s.data(), s.size()); out[s.size()] = 0; }
    return (int64_t) s.size();
}

int64_t orc_format_pixel_map(const uint8_t *rgb, int32_t rows, int32_t cols, uint8_t *out, size_t cap) {
    std::string s = formatPixelMap(rgb, rows, cols);
    if (out && cap >= s.size()) std::memcpy(out, s.data(), s.size());
    return (int64_t) s.size();
}
int64_t orc_parse_pixel_map(const uint8_t *data, size_t n, int32_t rows, int32_t cols, uint8_t *rgb, uint8_t *present) {
    if (present) std::memset(present, 0, (size_t) rows * (size_t) cols);
    return parsePixelMap(data, n, rows, cols, rgb, present);
}

// ---- per-function hooks (mirror the rt_dev_* hooks of include/rtfs_amd.h) --------------------------
int orc_float_producer(const uint32_t state[4], int32_t n, double *out) {
    FloatProducer p{state[0], state[1], state[2], state[3]};
    for (int i = 0; i < n; ++i) out[i] = p.Get();
    return RT_OK;
}
int orc_stream_state(uint64_t seed, int32_t n, const uint64_t *pixel, const uint32_t *sample, uint32_t *state_out) {
    for (int i = 0; i < n; ++i) {
        FloatProducer p = streamFor(seed, pixel[i], sample[i]);
        state_out[i * 4 + 0] = p.x; state_out[i * 4 + 1] = p.y; state_out[i * 4 + 2] = p.z; state_out[i * 4 + 3] = p.w;
    }
    return RT_OK;
}
static Ray rayOf(const double *r) { return Ray{P3(r), V3(r + 3)}; }

int orc_bbox_hits(int32_t n, const double *rays, const double *boxes, int32_t *hit) {
    for (int i = 0; i < n; ++i) {
        Ray r = rayOf(rays + i * 6);
        BoundingBox b{P3(boxes + i * 6), P3(boxes + i * 6 + 3)};
        hit[i] = BBox::hits(BBox::inverseDirections(r), r, b) ? 1 : 0;
    }
    return RT_OK;
}
int orc_sphere_first_intersection(int32_t n, const double *rays, const double *spheres, double *t) {
    SphereStyle st{RT_SPHERE_LIGHT_SOURCE, 1.0, 0, 0, 0, Colour::White, -1};
    for (int i = 0; i < n; ++i) {
        Sphere s = SphereM::make(st, P3(spheres + i * 4), spheres[i * 4 + 3]);
        double v;
        t[i] = SphereM::firstIntersection(s, rayOf(rays + i * 6), v) ? v : std::numeric_limits<double>::quiet_NaN();
    }
    return RT_OK;
}
int orc_plane_intersection(int32_t n, const double *rays, const double *planes, double *t) {
    for (int i = 0; i < n; ++i) {
        InfinitePlane p{InfinitePlaneStyle{RT_PLANE_LIGHT_SOURCE, 1.0, 0, Colour::White, -1}, V3(planes + i * 6 + 3), P3(planes + i * 6)};
        double v;
        t[i] = PlaneObj::intersection(p, rayOf(rays + i * 6), v) ? v : std::numeric_limits<double>::quiet_NaN();
    }
    return RT_OK;
}
int orc_pixel_combine(int32_t n, const uint8_t *a, const uint8_t *b, uint8_t *out) {
    for (int i = 0; i < n; ++i) {
        Pixel p = PixelM::combine(Pixel{a[i * 3], a[i * 3 + 1], a[i * 3 + 2]}, Pixel{b[i * 3], b[i * 3 + 1], b[i * 3 + 2]});
        out[i * 3] = p.Red; out[i * 3 + 1] = p.Green; out[i * 3 + 2] = p.Blue;
    }
    return RT_OK;
}
int orc_pixel_darken(int32_t n, const uint8_t *p, const double *albedo, uint8_t *out) {
    for (int i = 0; i < n; ++i) {
        Pixel q = PixelM::darken(albedo[i], Pixel{p[i * 3], p[i * 3 + 1], p[i * 3 + 2]});
        out[i * 3] = q.Red; out[i * 3 + 1] = q.Green; out[i * 3 + 2] = q.Blue;
    }
    return RT_OK;
}
int orc_reflection(const orc_scene *sc, int32_t n, const int32_t *index, const double *ray_in, const uint8_t *colour_in,
                   const double *strike, uint32_t *rng, int32_t *absorbed, uint8_t *colour_out, double *ray_out) {
    Counters cnt;
    for (int i = 0; i < n; ++i) {
        if (index[i] < 0 || (size_t) index[i] >= sc->s.objects.size()) { g_err = "index out of range"; return RT_ERR_INVALID_ARGUMENT; }
        LightRay lr{rayOf(ray_in + i * 6), Pixel{colour_in[i * 3], colour_in[i * 3 + 1], colour_in[i * 3 + 2]}};
        FloatProducer p{rng[i * 4], rng[i * 4 + 1], rng[i * 4 + 2], rng[i * 4 + 3]};
        Pixel col;
        bool ab = HittableM::reflection(sc->s.textures, sc->s.objects[(size_t) index[i]], lr, P3(strike + i * 3), p, col, cnt);
        absorbed[i] = ab ? 1 : 0;
        Pixel o = ab ? col : lr.Colour;
        colour_out[i * 3] = o.Red; colour_out[i * 3 + 1] = o.Green; colour_out[i * 3 + 2] = o.Blue;
        double *ro = ray_out + i * 6;
        ro[0] = lr.ray.Origin.x; ro[1] = lr.ray.Origin.y; ro[2] = lr.ray.Origin.z;
        ro[3] = lr.ray.Vector.x; ro[4] = lr.ray.Vector.y; ro[5] = lr.ray.Vector.z;
        rng[i * 4] = p.x; rng[i * 4 + 1] = p.y; rng[i * 4 + 2] = p.z; rng[i * 4 + 3] = p.w;
    }
    return RT_OK;
}
int orc_hit_object(const orc_scene *sc, int32_t n, const double *rays, int32_t *hit_index, double *strike, uint32_t *counters) {
    for (int i = 0; i < n; ++i) {
        Counters cnt;
        const Hittable *obj = nullptr;
        Point sp{std::numeric_limits<double>::quiet_NaN(), std::numeric_limits<double>::quiet_NaN(), std::numeric_limits<double>::quiet_NaN()};
        bool hit = SceneM::hitObject(sc->s, rayOf(rays + i * 6), obj, sp, cnt);
        hit_index[i] = hit ? obj->index : -1;
        if (strike) { strike[i * 3] = sp.x; strike[i * 3 + 1] = sp.y; strike[i * 3 + 2] = sp.z; }
        if (counters) { counters[i * 2] = (uint32_t) cnt.aabb; counters[i * 2 + 1] = (uint32_t) cnt.prim; }
    }
    return RT_OK;
}
int orc_trace_ray(const orc_scene *sc, int32_t bounce_depth, int32_t n, const double *rays, uint32_t *rng, uint8_t *colour_out) {
    Counters cnt;
    for (int i = 0; i < n; ++i) {
        LightRay lr{rayOf(rays + i * 6), Colour::White};
        FloatProducer p{rng[i * 4], rng[i * 4 + 1], rng[i * 4 + 2], rng[i * 4 + 3]};
        Pixel c = SceneM::traceRay(bounce_depth, sc->s, lr, p, cnt);
        colour_out[i * 3] = c.Red; colour_out[i * 3 + 1] = c.Green; colour_out[i * 3 + 2] = c.Blue;
        rng[i * 4] = p.x; rng[i * 4 + 1] = p.y; rng[i * 4 + 2] = p.z; rng[i * 4 + 3] = p.w;
    }
    return RT_OK;
}
int orc_texture_colour_at(const orc_scene *sc, int32_t texture, int32_t n, const double *points, double *uv_out, uint8_t *colour_out) {
    if (texture < 0 || (size_t) texture >= sc->s.textures.tex.size()) { g_err = "texture out of range"; return RT_ERR_INVALID_ARGUMENT; }
    for (int i = 0; i < n; ++i) {
        double uv[2] = {std::numeric_limits<double>::quiet_NaN(), std::numeric_limits<double>::quiet_NaN()};
        Pixel c = textureColourAt(sc->s.textures, texture, Colour::Black, P3(points + i * 3), uv);
        if (uv_out) { uv_out[i * 2] = uv[0]; uv_out[i * 2 + 1] = uv[1]; }
        colour_out[i * 3] = c.Red; colour_out[i * 3 + 1] = c.Green; colour_out[i * 3 + 2] = c.Blue;
    }
    return RT_OK;
}
int orc_arith(int32_t op, int32_t n, const double *a, const double *b, double *out) {
    for (int i = 0; i < n; ++i) {
        switch (op) {
        case 0: out[i] = 1.0 / a[i]; break;
        case 1: out[i] = std::sqrt(a[i]); break;
        case 2: out[i] = std::rint(a[i]); break;
        case 3: out[i] = a[i] / b[i]; break;
        case 4: out[i] = SphereM::pow5(a[i]); break;
        case 5: out[i] = std::pow(a[i], 5.0); break; // the C runtime's pow, for comparison
        case 7: out[i] = crAcos(a[i]); break;
        case 8: out[i] = crSin(a[i]); break;
        case 9: out[i] = crAtan2(a[i], b[i]); break;
        case 10: out[i] = std::acos(a[i]); break; // the C runtime's, for comparison
        case 11: out[i] = std::sin(a[i]); break;
        case 12: out[i] = std::atan2(a[i], b[i]); break;
        default: g_err = "bad op"; return RT_ERR_INVALID_ARGUMENT;
        }
    }
    return RT_OK;
}
// planeMap / planeMapInverse / liesOn / walkAlong / Plane.basis for the reference's property tests
int orc_plane_map(double radius, const double centre[3], double phi, double theta, double out[3]) {
    Point p = planeMap(radius, P3(centre), phi, theta);
    out[0] = p.x; out[1] = p.y; out[2] = p.z;
    return RT_OK;
}
int orc_plane_map_inverse(double radius, const double centre[3], const double p[3], double out[2]) {
    planeMapInverse(radius, P3(centre), P3(p), out[0], out[1]);
    return RT_OK;
}
int orc_sphere_lies_on(const double point[3], const double centre[3], double radius) {
    SphereStyle st{RT_SPHERE_LIGHT_SOURCE, 1.0, 0, 0, 0, Colour::White, -1};
    return SphereM::liesOn(P3(point), SphereM::make(st, P3(centre), radius)) ? 1 : 0;
}
int orc_ray_make(const double origin[3], const double vec[3], double out[6]) { // Ray.make'
    Ray r{};
    if (!RayM::makePrime(P3(origin), V3(vec), r)) return 0;
    out[0] = r.Origin.x; out[1] = r.Origin.y; out[2] = r.Origin.z; out[3] = r.Vector.x; out[4] = r.Vector.y; out[5] = r.Vector.z;
    return 1;
}
int orc_ray_walk_along(const double ray[6], double magnitude, double out[3]) {
    Point p = RayM::walkAlong(rayOf(ray), magnitude);
    out[0] = p.x; out[1] = p.y; out[2] = p.z;
    return RT_OK;
}
int orc_ray_lies_on(const double point[3], const double ray[6]) { return RayM::liesOn(P3(point), rayOf(ray)) ? 1 : 0; }
// Plane.orthonormalise + Plane.basis (Plane.fs:40-55, 82-97) as used by TestPlane.fs:12-26
int orc_plane_orthonormal_basis(const double origin[3], const double v1[3], const double v2[3], const double view_up[3],
                                double out_x[3], double out_y[3]) {
    OrthonormalPlane pl;
    if (!PlaneM::makeOrthonormalSpannedBy(Ray{P3(origin), V3(v1)}, Ray{P3(origin), V3(v2)}, pl)) return 0;
    Ray x, y;
    PlaneM::basis(V3(view_up), pl, x, y);
    out_x[0] = x.Vector.x; out_x[1] = x.Vector.y; out_x[2] = x.Vector.z;
    out_y[0] = y.Vector.x; out_y[1] = y.Vector.y; out_y[2] = y.Vector.z;
    return 1;
}
int orc_hardware_threads(void) { return (int) std::thread::hardware_concurrency(); }

} // extern "C"
