"""Small seeded scenes shared by the parity tests: every material, planes, textures, negative radii."""
import dataclasses

import numpy as np

import ray_tracing_fsharp_amd as rt
from ray_tracing_fsharp_amd import sample_images as si

P, V = rt.Point.make, rt.Vector.make
S, PS, H = rt.SphereStyle, rt.InfinitePlaneStyle, rt.Hittable
Tex = rt.Texture.Colour
Px = rt.Pixel


def unit(x, y, z):
    return rt.Vector.unitise(V(x, y, z))


def checker_image(h=16, w=32, seed=5):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)


def all_materials(spp=24, depth=12, pixels=20):
    """Every SphereStyle and InfinitePlaneStyle, bounded + unbounded, a hollow (negative radius) unbounded shell,
    a checkered UV-ramp texture, an image texture on a light; 16:9 like the reference's small scenes."""
    aspect = 16.0 / 9.0
    cam = dataclasses.replace(rt.Camera.makeBasic(spp, 1.0, aspect, P(0.0, 0.3, -1.0), unit(0.0, -0.05, 1.0), V(0.0, 1.0, 0.0)), BounceDepth=depth)
    chk = rt.ParameterisedTexture.Checkered(rt.ParameterisedTexture.UvRamp("u", 0, "v"), rt.ParameterisedTexture.UvRamp(100, "u", "v"), 50.0)
    img = rt.ParameterisedTexture.Image(checker_image())
    objs = [
        H.Sphere(rt.Sphere.make(S.LambertReflection(0.9, Tex(Px(25, 50, 120))), P(0.0, 0.0, 1.0), 0.5)),
        H.Sphere(rt.Sphere.make(S.PureReflection(0.95, rt.ParameterisedTexture.toTexture((0.5, P(1.1, 0.0, 1.0)), chk)), P(1.1, 0.0, 1.0), 0.5)),
        H.Sphere(rt.Sphere.make(S.Glass(0.95, Tex(rt.Colour.White), 1.5), P(-1.1, 0.0, 1.0), 0.5)),
        H.UnboundedSphere(rt.Sphere.make(S.Glass(1.0, Tex(rt.Colour.White), 1.0 / 1.5), P(-1.1, 0.0, 1.0), -0.4)),
        H.Sphere(rt.Sphere.make(S.FuzzedReflection(0.8, Tex(Px(255, 100, 0)), 0.3), P(0.5, 0.9, 1.6), 0.35)),
        H.Sphere(rt.Sphere.make(S.Dielectric(0.9, Tex(Px(200, 255, 200)), 1.4, 0.7), P(-0.5, 0.9, 1.6), 0.35)),
        H.Sphere(rt.Sphere.make(S.LightSourceCap(Px(255, 240, 200)), P(0.0, 1.9, 1.2), 0.4)),
        H.Sphere(rt.Sphere.make(S.LightSource(rt.ParameterisedTexture.toTexture((0.3, P(1.6, 1.2, 2.0)), img)), P(1.6, 1.2, 2.0), 0.3)),
        H.Sphere(rt.Sphere.make(S.Glass(1.0, Tex(rt.Colour.White), 1.5), P(0.2, -0.1, 0.2), -0.15)),  # bounded negative radius: never hit
        H.InfinitePlane(rt.InfinitePlane.make(PS.FuzzedReflection(0.85, Px(255, 200, 200), 0.4), P(0.0, -0.5, 0.0), unit(0.0, 1.0, 0.0))),
        H.InfinitePlane(rt.InfinitePlane.make(PS.PureReflection(0.8, rt.Colour.White), P(0.0, 0.0, 4.0), unit(0.3, 0.0, -1.0))),
        H.InfinitePlane(rt.InfinitePlane.make(PS.LambertReflection(0.7, Px(120, 220, 120)), P(-3.0, 0.0, 0.0), unit(1.0, 0.0, 0.0))),
        H.InfinitePlane(rt.InfinitePlane.make(PS.LightSource(Tex(Px(90, 90, 160))), P(0.0, 0.0, -5.0), unit(0.0, 0.0, 1.0))),
        H.UnboundedSphere(rt.Sphere.make(S.LightSource(Tex(Px(200, 200, 200))), P(0.0, 0.0, 0.0), 200.0)),
    ]
    return objs, cam, int(aspect * float(pixels)), pixels


def small_final(seed=7, spp=40, depth=50, pixels=16):
    """The RTIOW final scene (config 3) at thumbnail size: same geometry and tree, few pixels."""
    return si.config3_final(seed=seed, spp=spp, depth=depth, pixels=pixels)


def random_rays(n, seed, origin_scale=3.0):
    rng = np.random.default_rng(seed)
    o = rng.normal(size=(n, 3)) * origin_scale
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.concatenate([o, d], axis=1)


def earth_thumb(earthmap_rows_top_first, spp=30, pixels=18):
    """SampleImages.earth (SampleImages.fs:962-1010) at thumbnail size, with the committed decoded texels."""
    objs, cam, w, h = si.earth(earthmap_rows_top_first)
    cam = dataclasses.replace(cam, SamplesPerPixel=spp)
    aspect = 16.0 / 9.0
    return objs, cam, int(aspect * float(pixels)), pixels


def golden(name):
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"))


def many_spheres(n=2600, seed=11, spp=12, depth=8, pixels=10):
    """A scene whose flattened image (about 136 B per sphere) exceeds the 160 KiB LDS budget, so rt_render must take the
    global-memory variant of the kernel.  Mixed materials, a Lambert floor and a light dome."""
    rng = np.random.default_rng(seed)
    aspect = 16.0 / 9.0
    cam = dataclasses.replace(rt.Camera.makeBasic(spp, 1.0, aspect, P(0.0, 1.5, -6.0), unit(0.0, -0.2, 1.0), V(0.0, 1.0, 0.0)), BounceDepth=depth)
    objs = []
    for i in range(n):
        c = P(float(rng.uniform(-8, 8)), float(rng.uniform(0.1, 3.0)), float(rng.uniform(-2, 14)))
        r = float(rng.uniform(0.05, 0.3))
        col = Px(*(int(x) for x in rng.integers(30, 256, 3)))
        k = i % 5
        st = (S.LambertReflection(float(rng.uniform(0.3, 1.0)), Tex(col)) if k < 2 else
              S.FuzzedReflection(float(rng.uniform(0.5, 1.0)), Tex(col), float(rng.uniform(0.0, 0.5))) if k == 2 else
              S.Glass(1.0, Tex(rt.Colour.White), 1.5) if k == 3 else S.PureReflection(0.9, Tex(col)))
        objs.append(H.Sphere(rt.Sphere.make(st, c, r)))
    objs.append(H.UnboundedSphere(rt.Sphere.make(S.LambertReflection(0.5, Tex(Px(200, 200, 200))), P(0.0, -1000.0, 0.0), 1000.0)))
    objs.append(H.UnboundedSphere(rt.Sphere.make(S.LightSource(Tex(Px(220, 220, 255))), P(0.0, 0.0, 0.0), 3000.0)))
    return objs, cam, int(aspect * float(pixels)), pixels


def random_scene(seed, pixels=6):
    """A small random scene for fuzzing: every style can appear, bounded or unbounded, radii of both signs, planes in any
    orientation, camera anywhere (possibly inside objects), textures now and then."""
    rng = np.random.default_rng(seed)
    u = lambda a, b: float(rng.uniform(a, b))  # noqa: E731
    col = lambda: Px(*(int(x) for x in rng.integers(0, 256, 3)))  # noqa: E731
    chk = rt.ParameterisedTexture.Checkered(rt.ParameterisedTexture.UvRamp("u", int(rng.integers(0, 256)), "v"),
                                            rt.ParameterisedTexture.Colour(col()), u(5.0, 60.0))

    def tex(c, r):
        if rng.random() < 0.15:
            return rt.ParameterisedTexture.toTexture((abs(r), c), chk)
        return Tex(col())

    objs = []
    for _ in range(int(rng.integers(1, 9))):
        c = P(u(-2, 2), u(-1, 2), u(0, 5))
        r = u(0.1, 1.2) * (-1.0 if rng.random() < 0.15 else 1.0)
        k = int(rng.integers(0, 7))
        st = [S.LightSource(tex(c, r)), S.LightSourceCap(col()), S.PureReflection(u(0, 1), tex(c, r)), S.FuzzedReflection(u(0, 1), tex(c, r), u(0, 1)),
              S.LambertReflection(u(0, 1), tex(c, r)), S.Dielectric(u(0, 1), tex(c, r), u(0.7, 2.0), u(0, 1)), S.Glass(u(0.5, 1), tex(c, r), u(0.7, 2.0))][k]
        objs.append((H.UnboundedSphere if rng.random() < 0.3 else H.Sphere)(rt.Sphere.make(st, c, r)))
    for _ in range(int(rng.integers(0, 3))):
        n = rng.normal(size=3)
        k = int(rng.integers(0, 4))
        st = [PS.LightSource(Tex(col())), PS.PureReflection(u(0, 1), col()), PS.LambertReflection(u(0, 1), col()), PS.FuzzedReflection(u(0, 1), col(), u(0, 1))][k]
        objs.append(H.InfinitePlane(rt.InfinitePlane.make(st, P(u(-3, 3), u(-3, 3), u(-3, 6)), unit(*n))))
    if rng.random() < 0.7:
        objs.append(H.UnboundedSphere(rt.Sphere.make(S.LightSource(Tex(col())), P(0.0, 0.0, 0.0), u(20.0, 300.0))))
    rng.shuffle(objs)
    d = rng.normal(size=3)
    cam = dataclasses.replace(rt.Camera.makeBasic(int(rng.integers(1, 30)), u(0.5, 3.0), u(0.8, 2.0), P(u(-1, 1), u(-0.5, 1.5), u(-3, 1)), unit(*d), V(0.0, 1.0, 0.05)),
                              BounceDepth=int(rng.integers(0, 25)))
    return list(objs), cam, int(rng.integers(2, pixels + 1)), int(rng.integers(2, pixels + 1))
