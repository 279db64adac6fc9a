"""The reference's scene catalogue re-expressed as data through the host mirror (RayTracing.App/SampleImages.fs),
plus the benchmark instances SURVEY.md 8(d) defines (C2, C3).  Every function returns
(objects, camera, maxWidthCoord, maxHeightCoord); render with
`Scene.make objects |> Scene.render incr log maxW maxH camera` exactly as SampleImages.fs does.

The reference builds its random scene from unseeded `System.Random`s; here ONE FloatProducer stream seeded from
`seed` supplies every draw, in the order the reference's code makes them (documented per scene).
"""
from __future__ import annotations

import dataclasses
from typing import List, Tuple

import numpy as np

from . import _abi as A
from .raytracing import (Camera, Colour, FloatProducer, Hittable, InfinitePlane, InfinitePlaneStyle, ParameterisedTexture, Pixel,
                         Point, Sphere, SphereStyle, Texture, Vector, TOLERANCE)


def _mix64(z: int) -> int:
    z &= 0xFFFFFFFFFFFFFFFF
    z ^= z >> 30
    z = (z * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z ^= z >> 27
    z = (z * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    z ^= z >> 31
    return z


def scene_producer(seed: int) -> FloatProducer:
    """The host-side stream that replaces the scene's `Random ()` instances: four 31-bit words from SplitMix64(seed)."""
    a = _mix64(seed + 0x9E3779B97F4A7C15)
    b = _mix64(a + 0x9E3779B97F4A7C15)
    st = [(a & 0xFFFFFFFF) % 2147483647, (a >> 32) % 2147483647, (b & 0xFFFFFFFF) % 2147483647, (b >> 32) % 2147483647]
    if not any(st):
        st[3] = 1
    return FloatProducer(st)


def _unit(x, y, z) -> Vector:
    u = Vector.unitise(Vector.make(x, y, z))
    assert u is not None
    return u


def _extent(aspectRatio: float, pixels: int) -> Tuple[int, int]:
    # `aspectRatio * (float pixels) |> int` truncates (SampleImages.fs:96 and every scene)
    return int(aspectRatio * float(pixels)), pixels


_ORIGIN = Point.make(0.0, 0.0, 0.0)
_Z = _unit(0.0, 0.0, 1.0)
_UP = Vector.make(0.0, 1.0, 0.0)


def gradient() -> np.ndarray:
    """SampleImages.gradient (SampleImages.fs:37-57): 256x256 ramp, the renderer is not involved."""
    h, w = np.mgrid[0:256, 0:256]
    return np.stack([w.astype(np.uint8), (255 - h).astype(np.uint8), np.full((256, 256), 63, np.uint8)], axis=-1)


def shinyPlane():  # SampleImages.fs:59-96
    aspect = 16.0 / 9.0
    camera = Camera.makeBasic(50, 2.0, aspect, _ORIGIN, _Z, _UP)
    objs = [
        Hittable.Sphere(Sphere.make(SphereStyle.LightSource(Texture.Colour(Pixel(0, 255, 255))), Point.make(1.5, 0.5, 8.0), 0.5)),
        Hittable.InfinitePlane(InfinitePlane.make(InfinitePlaneStyle.PureReflection(0.5, Colour.White), Point.make(0.0, -1.0, 0.0), _unit(0.0, 1.0, 0.0))),
    ]
    return (objs, camera) + _extent(aspect, 400)


def fuzzyPlane():  # SampleImages.fs:98-136
    aspect = 16.0 / 9.0
    camera = Camera.makeBasic(50, 2.0, aspect, _ORIGIN, _Z, _UP)
    objs = [
        Hittable.Sphere(Sphere.make(SphereStyle.LightSource(Texture.Colour(Pixel(0, 255, 255))), Point.make(1.5, 0.5, 8.0), 0.5)),
        Hittable.InfinitePlane(InfinitePlane.make(InfinitePlaneStyle.FuzzedReflection(1.0, Colour.White, 0.75), Point.make(0.0, -1.0, 0.0), _unit(0.0, 1.0, 0.0))),
    ]
    return (objs, camera) + _extent(aspect, 400)


def spheres():  # SampleImages.fs:138-261
    aspect = 16.0 / 9.0
    camera = Camera.makeBasic(50, 7.0, aspect, _ORIGIN, _Z, _UP)
    objs = [
        Hittable.Sphere(Sphere.make(SphereStyle.LambertReflection(0.95, Texture.Colour(Pixel(255, 255, 0))), Point.make(0.0, 0.0, 9.0), 1.0)),
        Hittable.Sphere(Sphere.make(SphereStyle.PureReflection(1.0, Texture.Colour(Pixel(0, 255, 255))), Point.make(1.5, 0.5, 8.0), 0.5)),
        Hittable.Sphere(Sphere.make(SphereStyle.LightSource(Texture.Colour(Pixel(200, 220, 255))), Point.make(-1.5, 1.0, 8.0), 0.5)),
        Hittable.Sphere(Sphere.make(SphereStyle.FuzzedReflection(1.0, Texture.Colour(Pixel(255, 100, 0)), 0.2), Point.make(-0.4, 1.5, 10.0), 0.25)),
        Hittable.InfinitePlane(InfinitePlane.make(InfinitePlaneStyle.PureReflection(0.8, Colour.White), Point.make(0.0, 0.0, 12.0), _unit(1.0, 0.0, -1.0))),
        Hittable.InfinitePlane(InfinitePlane.make(InfinitePlaneStyle.FuzzedReflection(0.85, Pixel(255, 100, 100), 0.8), Point.make(0.0, -1.0, 0.0), _unit(0.0, 1.0, 0.0))),
        Hittable.InfinitePlane(InfinitePlane.make(InfinitePlaneStyle.PureReflection(0.95, Colour.White), Point.make(0.0, 0.0, 12.0), _unit(-1.0, 0.0, -1.0))),
        Hittable.InfinitePlane(InfinitePlane.make(InfinitePlaneStyle.LightSource(Texture.Colour(Pixel(15, 15, 15))), Point.make(0.0, 1.0, -1.0), _unit(0.0, 0.0, 1.0))),
    ]
    return (objs, camera) + _extent(aspect, 200)


def insideSphere():  # SampleImages.fs:263-411
    aspect = 16.0 / 9.0
    camera = Camera.makeBasic(50, 7.0, aspect, _ORIGIN, _Z, _UP)
    S, H = SphereStyle, Hittable
    objs = [
        H.Sphere(Sphere.make(S.LambertReflection(0.95, Texture.Colour(Pixel(255, 255, 0))), Point.make(0.0, 0.0, 9.0), 1.0)),
        H.Sphere(Sphere.make(S.PureReflection(1.0, Texture.Colour(Pixel(0, 255, 255))), Point.make(1.5, 0.5, 8.0), 0.5)),
        H.Sphere(Sphere.make(S.PureReflection(1.0, Texture.Colour(Pixel(255, 20, 20))), Point.make(-1.8, 0.8, 8.0), 0.5)),
        H.Sphere(Sphere.make(S.LightSource(Texture.Colour(Colour.White)), Point.make(-10.0, 8.0, 0.0), 9.0)),
        H.Sphere(Sphere.make(S.FuzzedReflection(1.0, Texture.Colour(Pixel(255, 100, 0)), 0.2), Point.make(1.4, 1.5, 10.0), 0.25)),
        H.Sphere(Sphere.make(S.PureReflection(0.9, Texture.Colour(Pixel(255, 255, 255))), Point.make(0.0, 10.0, 20.0), 8.0)),
        H.Sphere(Sphere.make(S.FuzzedReflection(0.6, Texture.Colour(Pixel(200, 50, 255)), 0.4), Point.make(0.0, -76.0, 9.0), 75.0)),
        H.Sphere(Sphere.make(S.FuzzedReflection(0.4, Texture.Colour(Pixel(200, 200, 200)), 0.0), Point.make(0.0, 0.0, 20.0), 100.0)),
        H.InfinitePlane(InfinitePlane.make(InfinitePlaneStyle.LightSource(Texture.Colour(Pixel(80, 80, 150))), Point.make(0.0, 0.0, -5.0), _unit(0.0, 0.0, 1.0))),
    ]
    return (objs, camera) + _extent(aspect, 1200)


def _three_spheres(floor_kind, right, middle, left, extra, light_kind, light_colour):
    S, H = SphereStyle, Hittable
    objs = [
        floor_kind(Sphere.make(S.LambertReflection(0.5, Texture.Colour(Pixel(204, 204, 0))), Point.make(0.0, -100.5, 1.0), 100.0)),
        H.Sphere(Sphere.make(right, Point.make(1.0, 0.0, 1.0), 0.5)),
        H.Sphere(Sphere.make(middle, Point.make(0.0, 0.0, 1.0), 0.5)),
        H.Sphere(Sphere.make(left, Point.make(-1.0, 0.0, 1.0), 0.5)),
    ]
    objs += extra
    objs.append(light_kind(Sphere.make(S.LightSource(Texture.Colour(light_colour)), Point.make(0.0, 0.0, 0.0), 200.0)))
    return objs


def totalRefraction():  # SampleImages.fs:413-503
    aspect = 16.0 / 9.0
    camera = Camera.makeBasic(50, 1.0, aspect, _ORIGIN, _Z, _UP)
    S = SphereStyle
    objs = _three_spheres(Hittable.Sphere, S.PureReflection(1.0, Texture.Colour(Pixel(204, 153, 51))),
                          S.LambertReflection(1.0, Texture.Colour(Pixel(25, 50, 120))),
                          S.Dielectric(1.0, Texture.Colour(Colour.White), 1.5, 1.0), [], Hittable.Sphere, Pixel(80, 80, 150))
    return (objs, camera) + _extent(aspect, 300)


def glassSphere():  # SampleImages.fs:505-597
    aspect = 16.0 / 9.0
    camera = Camera.makeBasic(50, 1.0, aspect, _ORIGIN, _Z, _UP)
    S = SphereStyle
    objs = _three_spheres(Hittable.UnboundedSphere, S.PureReflection(1.0, Texture.Colour(Pixel(100, 150, 200))),
                          S.LambertReflection(1.0, Texture.Colour(Pixel(25, 50, 120))),
                          S.Glass(0.9, Texture.Colour(Colour.White), 1.5), [], Hittable.UnboundedSphere, Pixel(200, 200, 200))
    return (objs, camera) + _extent(aspect, 200)


def texturedSphere():  # SampleImages.fs:599-700
    aspect = 16.0 / 9.0
    camera = Camera.makeBasic(50, 1.0, aspect, _ORIGIN, _Z, _UP)
    S = SphereStyle
    even = ParameterisedTexture.UvRamp("u", 0, "v")    # Red = byte (x*255), Green = 0, Blue = byte (y*255)
    odd = ParameterisedTexture.UvRamp(100, "u", "v")   # Red = 100, Green = byte (x*255), Blue = byte (y*255)
    texture = ParameterisedTexture.Checkered(even, odd, 50.0)
    right = S.PureReflection(1.0, ParameterisedTexture.toTexture((0.5, Point.make(1.0, 0.0, 1.0)), texture))
    objs = _three_spheres(Hittable.UnboundedSphere, right, S.LambertReflection(1.0, Texture.Colour(Pixel(25, 50, 120))),
                          S.Glass(0.9, Texture.Colour(Colour.White), 1.5), [], Hittable.UnboundedSphere, Pixel(200, 200, 200))
    return (objs, camera) + _extent(aspect, 200)


def movedCamera():  # SampleImages.fs:702-810 (the negative-radius sphere is BOUNDED there: never hit)
    aspect = 16.0 / 9.0
    origin = Point.make(-2.0, 2.0, -1.0)
    view = Vector.unitise(Point.differenceToThenFrom(Point.make(-1.0, 0.0, 1.0), origin))
    camera = Camera.makeBasic(50, 10.0, aspect, origin, view, _UP)
    S = SphereStyle
    shell = Hittable.Sphere(Sphere.make(S.Glass(1.0, Texture.Colour(Colour.White), 1.0 / 1.5), Point.make(-1.0, 0.0, 1.0), -0.45))
    objs = _three_spheres(Hittable.Sphere, S.PureReflection(1.0, Texture.Colour(Pixel(204, 153, 51))),
                          S.LambertReflection(1.0, Texture.Colour(Pixel(25, 50, 120))),
                          S.Glass(1.0, Texture.Colour(Colour.White), 1.5), [shell], Hittable.Sphere, Pixel(130, 130, 200))
    return (objs, camera) + _extent(aspect, 300)


def randomSpheres(seed: int = 2024, spp: int = 500, pixels: int = 800, grid: int = 11):
    """SampleImages.randomSpheres (SampleImages.fs:812-960), the RTIOW final scene = BASELINE config 3.
    `grid` is the reference's literal 11 (cells a, b in [-11, 10]); larger values give the same recipe over more cells -- scenes of
    ~4 * grid^2 spheres for the scene-size measurements (scripts/scene_sizes.py), never a BASELINE config.

    Draw order from the single scene stream, per grid cell (a, b) in the reference's loop order:
      materialChoice; centre.x offset; centre.z offset; then (if the cell is kept)
      Lambert: albedo factor 1, albedo factor 2, then 3 colour bytes (byte = int(Get()*256) clamped to 255);
      Fuzzed:  albedo draw, fuzz draw, then 3 colour bytes;   Glass: nothing.
    """
    rnd = scene_producer(seed)
    aspect = 3.0 / 2.0
    origin = Point.make(13.0, 2.0, -3.0)
    view = Vector.unitise(Point.differenceToThenFrom(Point.make(0.0, 0.0, 0.0), origin))
    camera = Camera.makeBasic(spp, 10.0, aspect, origin, view, _UP)
    S, H = SphereStyle, Hittable

    def colour_random() -> Pixel:  # Colour.random (Pixel.fs:68-76) draws 3 bytes from System.Random; ours come from the stream
        return Pixel(*(min(255, int(rnd.Get() * 256.0)) for _ in range(3)))

    def less(a: float, b: float) -> bool:  # Float.compare a b = Less
        return not (abs(a - b) < TOLERANCE) and a < b

    objs: List[Hittable] = []
    for a in range(-grid, grid):
        for b in range(-grid, grid):
            materialChoice = rnd.Get()
            centre = Point.make(float(a) + 0.9 * rnd.Get(), 0.2, float(b) + 0.9 * rnd.Get())
            d = Point.differenceToThenFrom(centre, Point.make(4.0, 0.2, 0.0))
            if Vector.dot(d, d) > 0.9 * 0.9:
                if less(materialChoice, 0.8):
                    albedo = rnd.Get() * rnd.Get() * 1.0
                    objs.append(H.Sphere(Sphere.make(S.LambertReflection(albedo, Texture.Colour(colour_random())), centre, 0.2)))
                elif less(materialChoice, 0.95):
                    albedo = rnd.Get() / 2.0 * 1.0 + 0.5
                    fuzz = rnd.Get() / 2.0 * 1.0
                    objs.append(H.Sphere(Sphere.make(S.FuzzedReflection(albedo, Texture.Colour(colour_random()), fuzz), centre, 0.2)))
                else:
                    objs.append(H.Sphere(Sphere.make(S.Glass(1.0, Texture.Colour(Colour.White), 1.5), centre, 0.2)))
    objs.append(H.Sphere(Sphere.make(S.Glass(1.0, Texture.Colour(Colour.White), 1.5), Point.make(0.0, 1.0, 0.0), 1.0)))
    objs.append(H.Sphere(Sphere.make(S.LambertReflection(1.0, Texture.Colour(Pixel(80, 40, 20))), Point.make(-4.0, 1.0, 0.0), 1.0)))
    objs.append(H.Sphere(Sphere.make(S.PureReflection(1.0, Texture.Colour(Pixel(180, 150, 128))), Point.make(4.0, 1.0, 0.0), 1.0)))
    objs.append(H.UnboundedSphere(Sphere.make(S.LightSource(Texture.Colour(Pixel(200, 200, 255))), Point.make(0.0, 0.0, 0.0), 2000.0)))  # ceiling
    objs.append(H.UnboundedSphere(Sphere.make(S.LambertReflection(0.5, Texture.Colour(Colour.White)), Point.make(0.0, -1000.0, 0.0), 1000.0)))  # floor
    return (objs, camera) + _extent(aspect, pixels)


def earth(earthmap_rows_top_first: np.ndarray):
    """SampleImages.earth (SampleImages.fs:962-1010); the decoded bitmap is passed in (no JPEG decode here)."""
    aspect = 16.0 / 9.0
    origin = Point.make(13.0, 2.0, -3.0)
    view = Vector.unitise(Point.differenceToThenFrom(Point.make(0.0, 0.0, 0.0), origin))
    camera = Camera.makeBasic(50, 12.0, aspect, origin, view, _UP)
    texture = ParameterisedTexture.ofImage(earthmap_rows_top_first)
    S, H = SphereStyle, Hittable
    objs = [
        H.Sphere(Sphere.make(S.LambertReflection(1.0, ParameterisedTexture.toTexture((1.0, Point.make(0.0, 0.0, 0.0)), texture)), Point.make(0.0, 0.0, 0.0), 1.0)),
        H.UnboundedSphere(Sphere.make(S.LightSource(Texture.Colour(Pixel(130, 130, 200))), Point.make(0.0, 0.0, 0.0), 200.0)),
    ]
    return (objs, camera) + _extent(aspect, 400)


# ---- BASELINE.json configs ----------------------------------------------------------------------------
def config1_empty():
    """C1 (ii): empty scene, maxW=100, maxH=50, 1 spp -> every pixel Black with exactly one sample (F2 in SURVEY.md)."""
    camera = Camera.makeBasic(1, 1.0, 2.0, _ORIGIN, _Z, _UP)
    return [], camera, 100, 50


def config2_three_lambert(spp: int = 100, depth: int = 50, pixels: int = 225):
    """C2: three Lambert spheres + Lambert floor + light dome, 16:9, depth override 50."""
    aspect = 16.0 / 9.0
    camera = dataclasses.replace(Camera.makeBasic(spp, 1.0, aspect, _ORIGIN, _Z, _UP), BounceDepth=depth)
    S, H = SphereStyle, Hittable
    objs = [
        H.Sphere(Sphere.make(S.LambertReflection(1.0, Texture.Colour(Pixel(25, 50, 120))), Point.make(-1.0, 0.0, 1.0), 0.5)),
        H.Sphere(Sphere.make(S.LambertReflection(0.5, Texture.Colour(Pixel(204, 204, 0))), Point.make(0.0, 0.0, 1.0), 0.5)),
        H.Sphere(Sphere.make(S.LambertReflection(0.8, Texture.Colour(Pixel(204, 153, 51))), Point.make(1.0, 0.0, 1.0), 0.5)),
        H.UnboundedSphere(Sphere.make(S.LambertReflection(0.5, Texture.Colour(Pixel(204, 204, 0))), Point.make(0.0, -100.5, 1.0), 100.0)),
        H.UnboundedSphere(Sphere.make(S.LightSource(Texture.Colour(Pixel(200, 200, 200))), Point.make(0.0, 0.0, 0.0), 200.0)),
    ]
    return (objs, camera) + _extent(aspect, pixels)


def config3_final(seed: int = 2024, spp: int = 500, depth: int = 50, pixels: int = 800):
    """C3: the final scene at BASELINE's 500 spp x 50 bounces; maxW=1200, maxH=800 -> 2401x1601 px (F5 in SURVEY.md)."""
    objs, camera, w, h = randomSpheres(seed, spp, pixels)
    return objs, dataclasses.replace(camera, BounceDepth=depth), w, h


def config5_mixed(earthmap_rows_top_first: np.ndarray, seed: int = 2024, spp: int = 2000, depth: int = 50, pixels: int = 800):
    """C5: final scene + image-textured Lambert sphere + one InfinitePlane mirror + a Dielectric sphere."""
    objs, camera, w, h = config3_final(seed, spp, depth, pixels)
    S, H = SphereStyle, Hittable
    tex = ParameterisedTexture.toTexture((1.0, Point.make(0.0, 1.0, 3.0)), ParameterisedTexture.ofImage(earthmap_rows_top_first))
    objs.append(H.Sphere(Sphere.make(S.LambertReflection(1.0, tex), Point.make(0.0, 1.0, 3.0), 1.0)))
    objs.append(H.Sphere(Sphere.make(S.Dielectric(1.0, Texture.Colour(Colour.White), 1.5, 0.9), Point.make(2.0, 0.6, 2.5), 0.6)))
    objs.append(H.InfinitePlane(InfinitePlane.make(InfinitePlaneStyle.PureReflection(0.8, Colour.White), Point.make(0.0, 0.0, -14.0), _unit(0.0, 0.0, 1.0))))
    return objs, camera, w, h


CATALOGUE = {  # SampleImages.Parse (SampleImages.fs:19-32); `gradient` and `earth` take data, see their functions
    "spheres": spheres, "shiny-floor": shinyPlane, "fuzzy-floor": fuzzyPlane, "inside-sphere": insideSphere,
    "total-refraction": totalRefraction, "moved-camera": movedCamera, "glass": glassSphere,
    "random-spheres": randomSpheres, "textured-sphere": texturedSphere,
}


def get(name: str):
    if name not in CATALOGUE:
        raise ValueError(f"Unrecognised arg: {name}")  # failwithf "Unrecognised arg: %s" (SampleImages.fs:32)
    return CATALOGUE[name]
