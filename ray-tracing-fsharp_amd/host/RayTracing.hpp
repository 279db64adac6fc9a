// RayTracing.hpp -- C++17 host-side mirror of the reference's scene-construction / render API over the C ABI
// (include/rtfs_amd.h).  Header-only; link with -lrtfs_amd.
//
// The reference is compiled F#; where its toolchain is absent this is the host layer a caller writes against.  Names,
// argument order and error behaviour follow the F# library (paths relative to /root/reference/RayTracing):
//   Point.make, Vector.unitise, Colour.*, Texture.Colour, ParameterisedTexture.*, SphereStyle.*, Sphere.make,
//   InfinitePlaneStyle.*, InfinitePlane.make, Hittable.*, Camera.makeBasic, Scene.make, Scene.render, Image,
//   ImageOutput.writePpm, FloatProducer.  F# `failwith` / ValueOption.get become std::runtime_error.
// Nothing here computes pixels: Scene::render hands the flattened scene to the HIP library when the Image is forced.
#pragma once

#include "../../include/rtfs_amd.h"

#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace RayTracing {

inline void check(int rc) {
    if (rc != RT_OK) throw std::runtime_error("rtfs_amd error " + std::to_string(rc) + ": " + rt_last_error());
}

// ---- Float.fs ------------------------------------------------------------------------------------------------------
namespace Float {
constexpr double tolerance = 0.00000001;                                             // Float.fs:82
inline bool equal(double a, double b) { return std::fabs(a - b) < tolerance; }       // Float.fs:84
inline bool less(double a, double b) { return !equal(a, b) && a < b; }               // Float.compare a b = Less
} // namespace Float

class FloatProducer { // Float.fs:14-47: xorshift128, byte reversal, / UInt32.MaxValue; host-side only (scene building)
  public:
    FloatProducer(uint32_t x, uint32_t y, uint32_t z, uint32_t w) : x_(x), y_(y), z_(z), w_(w) {}
    double Get() {
        uint32_t t = x_ ^ (x_ << 11);
        x_ = y_; y_ = z_; z_ = w_;
        w_ = w_ ^ (w_ >> 19) ^ (t ^ (t >> 8));
        uint32_t i = ((w_ & 0xFFu) << 24) ^ (((w_ >> 8) & 0xFFu) << 16) ^ (((w_ >> 16) & 0xFFu) << 8) ^ ((w_ >> 24) & 0xFFu);
        return (double) i / (double) 4294967295u;
    }
  private:
    uint32_t x_, y_, z_, w_;
};

// ---- Point.fs --------------------------------------------------------------------------------------------------------
struct Vector {
    double x, y, z;
    static Vector make(double x, double y, double z) { return Vector{x, y, z}; }
    static double dot(Vector a, Vector b) { return a.x * b.x + a.y * b.y + a.z * b.z; }                // Point.fs:18
    static std::optional<Vector> unitise(Vector v) {                                                    // Point.fs:28-35
        double d = dot(v, v);
        if (Float::equal(d, 0.0)) return std::nullopt;
        double f = 1.0 / std::sqrt(d);
        return Vector{f * v.x, f * v.y, f * v.z};
    }
};
using UnitVector = Vector;
struct Point {
    double x, y, z;
    static Point make(double x, double y, double z) { return Point{x, y, z}; }
    static Vector differenceToThenFrom(Point p, Point q) { return Vector{p.x - q.x, p.y - q.y, p.z - q.z}; } // Point.fs:94
};
inline UnitVector unit(double x, double y, double z) { // `Vector.make x y z |> Vector.unitise |> ValueOption.get`
    auto u = Vector::unitise(Vector::make(x, y, z));
    if (!u) throw std::runtime_error("ValueOption.get: cannot unitise a zero vector");
    return *u;
}

// ---- Pixel.fs --------------------------------------------------------------------------------------------------------
struct Pixel { uint8_t Red, Green, Blue; };
namespace Colour {                                   // Pixel.fs:18-66
constexpr Pixel Black{0, 0, 0}, White{255, 255, 255}, Red{255, 0, 0}, Green{0, 255, 0}, Blue{0, 0, 255}, Yellow{255, 255, 0},
    HotPink{205, 105, 180};
}

// ---- Texture.fs ------------------------------------------------------------------------------------------------------
struct ParameterisedTexture { // Texture.fs:19-24; Arbitrary closures are enumerated (UvRamp), see include/rtfs_amd.h
    uint32_t kind = RT_TEXTURE_COLOUR;
    Pixel pixel = Colour::Black;
    std::shared_ptr<ParameterisedTexture> even, odd;
    double gridSize = 0.0;
    int width = 0, height = 0;
    std::shared_ptr<std::vector<uint8_t>> texels; // img.[y].[x] order
    uint8_t ramp[3] = {RT_RAMP_CONST, RT_RAMP_CONST, RT_RAMP_CONST};

    static ParameterisedTexture Colour(Pixel p) { ParameterisedTexture t; t.pixel = p; return t; }
    static ParameterisedTexture Checkered(ParameterisedTexture e, ParameterisedTexture o, double grid) {
        ParameterisedTexture t; t.kind = RT_TEXTURE_CHECKERED; t.gridSize = grid;
        t.even = std::make_shared<ParameterisedTexture>(std::move(e)); t.odd = std::make_shared<ParameterisedTexture>(std::move(o));
        return t;
    }
    static ParameterisedTexture Image(std::vector<uint8_t> rows, int width, int height) {
        ParameterisedTexture t; t.kind = RT_TEXTURE_IMAGE; t.width = width; t.height = height;
        t.texels = std::make_shared<std::vector<uint8_t>>(std::move(rows));
        return t;
    }
    // ParameterisedTexture.ofImage (Texture.fs:30-48): bitmap rows top-first in, reversed rows stored
    static ParameterisedTexture ofImage(const std::vector<uint8_t> &bitmapTopFirst, int width, int height) {
        std::vector<uint8_t> rows(bitmapTopFirst.size());
        const size_t stride = (size_t) width * 3;
        for (int y = 0; y < height; ++y) std::memcpy(&rows[(size_t) y * stride], &bitmapTopFirst[(size_t) (height - y - 1) * stride], stride);
        return Image(std::move(rows), width, height);
    }
    // channel = constant byte, or 'u' (byte (x * 255.0)) / 'v' (byte (y * 255.0)): RayTracing.App/SampleImages.fs:606-627
    static ParameterisedTexture UvRamp(int red, int green, int blue) {
        ParameterisedTexture t; t.kind = RT_TEXTURE_UV_RAMP;
        const int ch[3] = {red, green, blue};
        uint8_t c[3] = {0, 0, 0};
        for (int k = 0; k < 3; ++k) {
            if (ch[k] == 'u') t.ramp[k] = RT_RAMP_U;
            else if (ch[k] == 'v') t.ramp[k] = RT_RAMP_V;
            else c[k] = (uint8_t) ch[k];
        }
        t.pixel = Pixel{c[0], c[1], c[2]};
        return t;
    }
};
constexpr int U = 'u', V = 'v';

struct Texture { // Texture.fs:6-8
    bool isColour = true;
    Pixel pixel = Colour::Black;
    ParameterisedTexture param;
    double mapRadius = 1.0;
    Point mapCentre{0, 0, 0};
    static Texture Colour(Pixel p) { Texture t; t.pixel = p; return t; }
};
// ParameterisedTexture.toTexture (Sphere.planeMapInverse radius centre) texture (Texture.fs:69-72)
inline Texture toTexture(double radius, Point centre, ParameterisedTexture tex) {
    Texture t; t.isColour = false; t.param = std::move(tex); t.mapRadius = radius; t.mapCentre = centre; return t;
}

// ---- Sphere.fs / InfinitePlane.fs / Hittable.fs ---------------------------------------------------------------------------
struct Style {
    uint32_t style = 0;
    double albedo = 1.0, fuzz = 0.0, ior = 1.0, prob = 0.0;
    Pixel colour = Colour::Black;
    std::optional<Texture> texture;
};
namespace SphereStyle { // Sphere.fs:10-37 (the FloatProducer members do not cross the ABI: randomness comes from the seed)
inline Style LightSource(Texture t) { Style s; s.style = RT_SPHERE_LIGHT_SOURCE; s.texture = std::move(t); return s; }
inline Style LightSourceCap(Pixel p) { Style s; s.style = RT_SPHERE_LIGHT_SOURCE_CAP; s.colour = p; return s; }
inline Style PureReflection(double albedo, Texture t) { Style s; s.style = RT_SPHERE_PURE_REFLECTION; s.albedo = albedo; s.texture = std::move(t); return s; }
inline Style FuzzedReflection(double albedo, Texture t, double fuzz) { Style s; s.style = RT_SPHERE_FUZZED_REFLECTION; s.albedo = albedo; s.fuzz = fuzz; s.texture = std::move(t); return s; }
inline Style LambertReflection(double albedo, Texture t) { Style s; s.style = RT_SPHERE_LAMBERT_REFLECTION; s.albedo = albedo; s.texture = std::move(t); return s; }
inline Style Dielectric(double albedo, Texture t, double ior, double prob) { Style s; s.style = RT_SPHERE_DIELECTRIC; s.albedo = albedo; s.ior = ior; s.prob = prob; s.texture = std::move(t); return s; }
inline Style Glass(double albedo, Texture t, double ior) { Style s; s.style = RT_SPHERE_GLASS; s.albedo = albedo; s.ior = ior; s.texture = std::move(t); return s; }
} // namespace SphereStyle
namespace InfinitePlaneStyle { // InfinitePlane.fs:3-13
inline Style LightSource(Texture t) { Style s; s.style = RT_PLANE_LIGHT_SOURCE; s.texture = std::move(t); return s; }
inline Style PureReflection(double albedo, Pixel c) { Style s; s.style = RT_PLANE_PURE_REFLECTION; s.albedo = albedo; s.colour = c; return s; }
inline Style LambertReflection(double albedo, Pixel c) { Style s; s.style = RT_PLANE_LAMBERT_REFLECTION; s.albedo = albedo; s.colour = c; return s; }
inline Style FuzzedReflection(double albedo, Pixel c, double fuzz) { Style s; s.style = RT_PLANE_FUZZED_REFLECTION; s.albedo = albedo; s.colour = c; s.fuzz = fuzz; return s; }
} // namespace InfinitePlaneStyle

struct Sphere { Style style; Point Centre; double Radius; static Sphere make(Style s, Point c, double r) { return Sphere{std::move(s), c, r}; } };            // Sphere.fs:325
struct InfinitePlane { Style style; UnitVector Normal; Point P; static InfinitePlane make(Style s, Point p, UnitVector n) { return InfinitePlane{std::move(s), n, p}; } }; // InfinitePlane.fs:114

struct Hittable { // Hittable.fs:3-6
    uint32_t kind;
    Style style;
    Point point{0, 0, 0};
    Vector normal{0, 0, 0};
    double radius = 0.0;
    static Hittable Sphere(const RayTracing::Sphere &s) { return Hittable{RT_HITTABLE_SPHERE, s.style, s.Centre, {0, 0, 0}, s.Radius}; }
    static Hittable UnboundedSphere(const RayTracing::Sphere &s) { return Hittable{RT_HITTABLE_UNBOUNDED_SPHERE, s.style, s.Centre, {0, 0, 0}, s.Radius}; }
    static Hittable InfinitePlane(const RayTracing::InfinitePlane &p) { return Hittable{RT_HITTABLE_INFINITE_PLANE, p.style, p.P, p.Normal, 0.0}; }
};

// ---- Camera.fs -------------------------------------------------------------------------------------------------------
struct Camera { // Camera.fs:3-28; fields are public as in the reference, e.g. `camera.BounceDepth = 50`
    rt_camera abi{};
    int SamplesPerPixel = 1;
    int BounceDepth = 150;
    static Camera makeBasic(int samplesPerPixel, double focalLength, double aspectRatio, Point origin, UnitVector viewDirection, Vector viewUp) {
        Camera c;
        const double o[3] = {origin.x, origin.y, origin.z}, d[3] = {viewDirection.x, viewDirection.y, viewDirection.z}, up[3] = {viewUp.x, viewUp.y, viewUp.z};
        check(rt_camera_make_basic(samplesPerPixel, focalLength, aspectRatio, o, d, up, &c.abi));
        c.SamplesPerPixel = samplesPerPixel;
        c.BounceDepth = c.abi.bounce_depth;
        return c;
    }
    rt_camera toAbi() const { rt_camera a = abi; a.samples_per_pixel = SamplesPerPixel; a.bounce_depth = BounceDepth; return a; }
};

// ---- Domain.fs ---------------------------------------------------------------------------------------------------------
class Image { // Domain.fs:9-31: rows are produced on first use
  public:
    Image(int rows, int cols, std::function<std::vector<uint8_t>()> force) : RowCount(rows), ColCount(cols), force_(std::move(force)) {}
    int RowCount, ColCount;
    const std::vector<uint8_t> &render() { // Image.render: [RowCount][ColCount][3], row 0 = top
        if (!rows_) rows_ = std::make_shared<std::vector<uint8_t>>(force_());
        return *rows_;
    }
  private:
    std::function<std::vector<uint8_t>()> force_;
    std::shared_ptr<std::vector<uint8_t>> rows_;
};

// ---- Scene.fs ----------------------------------------------------------------------------------------------------------
class Scene {
  public:
    ~Scene() { if (h_) rt_scene_destroy(h_); }
    Scene(const Scene &) = delete;
    Scene &operator=(const Scene &) = delete;

    static std::shared_ptr<Scene> make(const std::vector<Hittable> &objects) { // Scene.fs:15-28
        std::vector<rt_hittable> hs(objects.size());
        std::vector<rt_texture> texs;
        std::vector<std::shared_ptr<std::vector<uint8_t>>> keep;
        std::function<int(const ParameterisedTexture &)> add = [&](const ParameterisedTexture &t) -> int {
            rt_texture r{};
            r.kind = t.kind; r.even = r.odd = -1; r.map_radius = 1.0;
            if (t.kind == RT_TEXTURE_CHECKERED) { const int e = add(*t.even), o = add(*t.odd); r.even = e; r.odd = o; r.grid_size = t.gridSize; } // children first
            else if (t.kind == RT_TEXTURE_IMAGE) { keep.push_back(t.texels); r.width = t.width; r.height = t.height; r.texels = t.texels->data(); }
            else if (t.kind == RT_TEXTURE_UV_RAMP) std::memcpy(r.ramp_src, t.ramp, 3);
            r.rgb[0] = t.pixel.Red; r.rgb[1] = t.pixel.Green; r.rgb[2] = t.pixel.Blue;
            texs.push_back(r);
            return (int) texs.size() - 1;
        };
        for (size_t i = 0; i < objects.size(); ++i) {
            const Hittable &h = objects[i];
            rt_hittable &o = hs[i];
            std::memset(&o, 0, sizeof(o));
            o.kind = h.kind; o.style = h.style.style; o.texture = -1;
            o.albedo = h.style.albedo; o.fuzz = h.style.fuzz; o.ior = h.style.ior; o.prob = h.style.prob;
            Pixel colour = h.style.colour;
            if (h.style.texture) {
                if (h.style.texture->isColour) colour = h.style.texture->pixel;
                else {
                    const int idx = add(h.style.texture->param);
                    texs[(size_t) idx].map_radius = h.style.texture->mapRadius;
                    texs[(size_t) idx].map_centre[0] = h.style.texture->mapCentre.x; texs[(size_t) idx].map_centre[1] = h.style.texture->mapCentre.y;
                    texs[(size_t) idx].map_centre[2] = h.style.texture->mapCentre.z;
                    o.texture = idx;
                }
            }
            o.rgb[0] = colour.Red; o.rgb[1] = colour.Green; o.rgb[2] = colour.Blue;
            o.point[0] = h.point.x; o.point[1] = h.point.y; o.point[2] = h.point.z;
            o.normal[0] = h.normal.x; o.normal[1] = h.normal.y; o.normal[2] = h.normal.z;
            o.radius = h.radius;
        }
        rt_scene *raw = nullptr;
        check(rt_scene_create(hs.data(), hs.size(), texs.empty() ? nullptr : texs.data(), texs.size(), &raw));
        return std::shared_ptr<Scene>(new Scene(raw));
    }

    // Scene.render progressIncrement log maxWidthCoord maxHeightCoord camera scene (Scene.fs:196-236): returns
    // (rows as progress units, lazy Image); the work happens when the Image is forced, then progress ticks once per row.
    static std::pair<double, Image> render(std::function<void(double)> progressIncrement, std::function<void(const std::string &)> /*log*/,
                                           int maxWidthCoord, int maxHeightCoord, const Camera &camera, std::shared_ptr<Scene> s,
                                           uint64_t seed = 0, int device = 0) {
        const int rowsIter = 2 * maxHeightCoord + 1, colsIter = 2 * maxWidthCoord + 1;
        auto force = [=]() {
            std::vector<int32_t> accum((size_t) rowsIter * (size_t) colsIter * 4);
            std::vector<uint8_t> rgb((size_t) rowsIter * (size_t) colsIter * 3);
            rt_camera cam = camera.toAbi();
            check(rt_render(s->h_, &cam, maxWidthCoord, maxHeightCoord, seed, device, 0, 1, rowsIter, 0u, accum.data(), rgb.data(), &s->lastStats));
            for (int r = 0; r < rowsIter; ++r) progressIncrement(1.0);
            return rgb;
        };
        return {(double) rowsIter, Image(rowsIter, colsIter, force)};
    }

    // rt_scene_tune (no counterpart in the reference): the walk tree rebuilt from the rays of a small probe render with this camera;
    // same pixels, about a third fewer box tests per ray, ~12 % off the frame time.  Costs ~20 ms: worth it from a scene's second frame on, or
    // for a first frame of more than about 0.15 s.
    rt_tune_info tune(int maxWidthCoord, int maxHeightCoord, const Camera &camera, uint64_t seed = 0, int device = 0) {
        rt_tune_info info{};
        info.struct_size = sizeof(info);
        rt_camera cam = camera.toAbi();
        check(rt_scene_tune(h_, &cam, maxWidthCoord, maxHeightCoord, seed, device, &info));
        return info;
    }

    rt_scene *handle() const { return h_; }
    rt_stats lastStats{};

  private:
    explicit Scene(rt_scene *h) : h_(h) {}
    rt_scene *h_;
};

// ---- ImageOutput.fs -----------------------------------------------------------------------------------------------------
namespace PixelOutput { inline uint8_t correct(uint8_t b) { return rt_gamma_correct(b); } } // ImageOutput.fs:11-18
namespace ImageOutput {
inline void writePpm(bool gammaCorrect, const std::function<void(double)> &incrementProgress, const std::vector<uint8_t> &pixels, int rows, int cols,
                     const std::string &output) { // ImageOutput.fs:163-197
    check(rt_write_ppm(output.c_str(), pixels.data(), rows, cols, gammaCorrect ? 1 : 0));
    for (long i = 0; i < (long) rows * cols; ++i) incrementProgress(1.0);
}
} // namespace ImageOutput

} // namespace RayTracing
