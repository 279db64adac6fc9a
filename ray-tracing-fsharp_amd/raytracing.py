"""Host-side mirror of the reference's scene-construction / render API over the C ABI (include/rtfs_amd.h).

Same names, argument order and error behaviour as the F# library (paths relative to /root/reference/RayTracing):
`Point.make`, `Vector.unitise`, `Colour.*`, `Texture.Colour`, `ParameterisedTexture.*`, `SphereStyle.*`, `Sphere.make`,
`InfinitePlaneStyle.*`, `InfinitePlane.make`, `Hittable.*`, `Camera.makeBasic`, `Scene.make`, `Scene.render`,
`Image.render`, `ImageOutput.writePpm`, `PixelOutput.correct`, `FloatProducer`.

Nothing here computes pixels: `Scene.render` hands the flattened scene to the HIP library.  Two things cannot cross
a C ABI and are mapped instead: the `FloatProducer` arguments of the styles are accepted and ignored
(randomness comes from `seed`), and `Texture.Arbitrary` closures must be one of the enumerated texture kinds.
"""
from __future__ import annotations

import ctypes as C
import dataclasses
import math
from typing import Callable, List, NamedTuple, Optional, Sequence, Tuple

import numpy as np

from . import _abi as A
from ._lib import RtError, check, lib

TOLERANCE = 0.00000001  # Float.fs:82


# ---- Float.fs:31-76 --------------------------------------------------------------------------------------------
class FloatProducer:
    """xorshift128 + byte reversal + /UInt32.MaxValue (Float.fs:14-47).  `seed` is anything with four
    `.Next()`-style draws or an explicit (x, y, z, w); used host-side only, to build scenes reproducibly."""

    def __init__(self, state: Sequence[int]):
        self.x, self.y, self.z, self.w = (int(v) & 0xFFFFFFFF for v in state)

    def _next(self) -> int:
        t = (self.x ^ ((self.x << 11) & 0xFFFFFFFF)) & 0xFFFFFFFF
        self.x, self.y, self.z = self.y, self.z, self.w
        self.w = (self.w ^ (self.w >> 19) ^ (t ^ (t >> 8))) & 0xFFFFFFFF
        return self.w

    def Get(self) -> float:
        w = self._next()
        i = ((w & 0xFF) << 24) ^ (((w >> 8) & 0xFF) << 16) ^ (((w >> 16) & 0xFF) << 8) ^ ((w >> 24) & 0xFF)
        return float(i) / float(0xFFFFFFFF)

    def GetTwo(self) -> Tuple[float, float]:
        return self.Get(), self.Get()

    def GetThree(self) -> Tuple[float, float, float]:
        return self.Get(), self.Get(), self.Get()


# ---- Point.fs -----------------------------------------------------------------------------------------------------
class Point(NamedTuple):
    x: float
    y: float
    z: float

    @staticmethod
    def make(x: float, y: float, z: float) -> "Point":
        return Point(float(x), float(y), float(z))

    @staticmethod
    def differenceToThenFrom(p: "Point", q: "Point") -> "Vector":  # Point.fs:94
        return Vector(p.x - q.x, p.y - q.y, p.z - q.z)


class Vector(NamedTuple):
    x: float
    y: float
    z: float

    @staticmethod
    def make(x: float, y: float, z: float) -> "Vector":
        return Vector(float(x), float(y), float(z))

    @staticmethod
    def dot(a: "Vector", b: "Vector") -> float:  # Point.fs:18
        return a.x * b.x + a.y * b.y + a.z * b.z

    @staticmethod
    def unitise(v: "Vector") -> Optional["Vector"]:  # Point.fs:28-35 -> UnitVector voption
        d = Vector.dot(v, v)
        if abs(d - 0.0) < TOLERANCE:
            return None
        f = 1.0 / math.sqrt(d)
        return Vector(f * v.x, f * v.y, f * v.z)


UnitVector = Vector


# ---- Pixel.fs -----------------------------------------------------------------------------------------------------
class Pixel(NamedTuple):
    Red: int
    Green: int
    Blue: int


class Colour:  # Pixel.fs:18-76
    Black = Pixel(0, 0, 0)
    White = Pixel(255, 255, 255)
    Red = Pixel(255, 0, 0)
    Green = Pixel(0, 255, 0)
    Blue = Pixel(0, 0, 255)
    Yellow = Pixel(255, 255, 0)
    HotPink = Pixel(205, 105, 180)


# ---- Texture.fs ---------------------------------------------------------------------------------------------------
@dataclasses.dataclass(frozen=True)
class _ParamTex:
    kind: int
    pixel: Pixel = Colour.Black
    even: Optional["_ParamTex"] = None
    odd: Optional["_ParamTex"] = None
    grid: float = 0.0
    image: Optional[np.ndarray] = None  # [H, W, 3] uint8, img.[y].[x] order (Texture.fs:63-67)
    ramp: Tuple[int, int, int] = (0, 0, 0)


class ParameterisedTexture:  # Texture.fs:19-24
    @staticmethod
    def Colour(p: Pixel) -> _ParamTex:
        return _ParamTex(A.RT_TEXTURE_COLOUR, pixel=Pixel(*p))

    @staticmethod
    def Checkered(even: _ParamTex, odd: _ParamTex, gridSize: float) -> _ParamTex:
        return _ParamTex(A.RT_TEXTURE_CHECKERED, even=even, odd=odd, grid=float(gridSize))

    @staticmethod
    def Image(rows: np.ndarray) -> _ParamTex:
        img = np.ascontiguousarray(rows, dtype=np.uint8)
        if img.ndim != 3 or img.shape[2] != 3:
            raise ValueError("Image texture must be [height, width, 3] uint8")
        return _ParamTex(A.RT_TEXTURE_IMAGE, image=img)

    @staticmethod
    def ofImage(bitmap_rows_top_first: np.ndarray) -> _ParamTex:
        """ParameterisedTexture.ofImage (Texture.fs:30-48): rows are reversed (y = Height - y - 1)."""
        return ParameterisedTexture.Image(np.ascontiguousarray(bitmap_rows_top_first[::-1]))

    @staticmethod
    def UvRamp(red, green, blue) -> _ParamTex:
        """The closed forms of the `ParameterisedTexture.Arbitrary` closures in RayTracing.App/SampleImages.fs:606-627:
        each channel is a constant byte, "u" (= byte (x * 255.0)) or "v" (= byte (y * 255.0))."""
        src, const = [], []
        for ch in (red, green, blue):
            if ch == "u":
                src.append(A.RT_RAMP_U); const.append(0)
            elif ch == "v":
                src.append(A.RT_RAMP_V); const.append(0)
            else:
                src.append(A.RT_RAMP_CONST); const.append(int(ch))
        return _ParamTex(A.RT_TEXTURE_UV_RAMP, pixel=Pixel(*const), ramp=tuple(src))

    @staticmethod
    def Arbitrary(_f: Callable) -> _ParamTex:
        raise RtError(A.RT_ERR_UNSUPPORTED, "Texture.Arbitrary closures cannot cross the C ABI; use UvRamp/Checkered/Image")

    @staticmethod
    def toTexture(interpret: Tuple[float, Point], texture: _ParamTex) -> "Texture":
        """ParameterisedTexture.toTexture (Texture.fs:69-72).  `interpret` is the pair (radius, centre) that the
        reference passes as `Sphere.planeMapInverse radius centre`."""
        radius, centre = interpret
        return Texture(param=texture, map_radius=float(radius), map_centre=Point(*centre))


@dataclasses.dataclass(frozen=True)
class Texture:  # Texture.fs:6-8
    pixel: Optional[Pixel] = None
    param: Optional[_ParamTex] = None
    map_radius: float = 1.0
    map_centre: Point = Point(0.0, 0.0, 0.0)

    @staticmethod
    def Colour(p: Pixel) -> "Texture":
        return Texture(pixel=Pixel(*p))


# ---- Sphere.fs / InfinitePlane.fs / Hittable.fs ------------------------------------------------------------------------
@dataclasses.dataclass(frozen=True)
class _Style:
    style: int
    albedo: float = 1.0
    texture: Optional[Texture] = None
    colour: Pixel = Colour.Black
    fuzz: float = 0.0
    ior: float = 1.0
    prob: float = 0.0


class SphereStyle:  # Sphere.fs:10-37
    @staticmethod
    def LightSource(texture: Texture) -> _Style:
        return _Style(A.RT_SPHERE_LIGHT_SOURCE, texture=texture)

    @staticmethod
    def LightSourceCap(colour: Pixel) -> _Style:
        return _Style(A.RT_SPHERE_LIGHT_SOURCE_CAP, colour=Pixel(*colour))

    @staticmethod
    def PureReflection(albedo: float, texture: Texture) -> _Style:
        return _Style(A.RT_SPHERE_PURE_REFLECTION, albedo=float(albedo), texture=texture)

    @staticmethod
    def FuzzedReflection(albedo: float, texture: Texture, fuzz: float, rand=None) -> _Style:
        return _Style(A.RT_SPHERE_FUZZED_REFLECTION, albedo=float(albedo), texture=texture, fuzz=float(fuzz))

    @staticmethod
    def LambertReflection(albedo: float, texture: Texture, rand=None) -> _Style:
        return _Style(A.RT_SPHERE_LAMBERT_REFLECTION, albedo=float(albedo), texture=texture)

    @staticmethod
    def Dielectric(albedo: float, texture: Texture, boundaryRefractance: float, refraction: float, rand=None) -> _Style:
        return _Style(A.RT_SPHERE_DIELECTRIC, albedo=float(albedo), texture=texture, ior=float(boundaryRefractance),
                      prob=float(refraction))

    @staticmethod
    def Glass(albedo: float, texture: Texture, ior: float, rand=None) -> _Style:
        return _Style(A.RT_SPHERE_GLASS, albedo=float(albedo), texture=texture, ior=float(ior))


class InfinitePlaneStyle:  # InfinitePlane.fs:3-13
    @staticmethod
    def LightSource(texture: Texture) -> _Style:
        return _Style(A.RT_PLANE_LIGHT_SOURCE, texture=texture)

    @staticmethod
    def PureReflection(albedo: float, colour: Pixel) -> _Style:
        return _Style(A.RT_PLANE_PURE_REFLECTION, albedo=float(albedo), colour=Pixel(*colour))

    @staticmethod
    def LambertReflection(albedo: float, colour: Pixel, rand=None) -> _Style:
        return _Style(A.RT_PLANE_LAMBERT_REFLECTION, albedo=float(albedo), colour=Pixel(*colour))

    @staticmethod
    def FuzzedReflection(albedo: float, colour: Pixel, fuzz: float, rand=None) -> _Style:
        return _Style(A.RT_PLANE_FUZZED_REFLECTION, albedo=float(albedo), colour=Pixel(*colour), fuzz=float(fuzz))


@dataclasses.dataclass(frozen=True)
class Sphere:  # Sphere.fs:302-337
    Style: _Style
    Centre: Point
    Radius: float

    @staticmethod
    def make(style: _Style, centre: Point, radius: float) -> "Sphere":
        return Sphere(style, Point(*centre), float(radius))


@dataclasses.dataclass(frozen=True)
class InfinitePlane:  # InfinitePlane.fs:101-119
    Style: _Style
    Normal: UnitVector
    Point: Point

    @staticmethod
    def make(style: _Style, pointOnPlane: Point, normal: UnitVector) -> "InfinitePlane":
        return InfinitePlane(style, Vector(*normal), Point(*pointOnPlane))


@dataclasses.dataclass(frozen=True)
class Hittable:  # Hittable.fs:3-6
    kind: int
    sphere: Optional[Sphere] = None
    plane: Optional[InfinitePlane] = None

    @staticmethod
    def Sphere(s: Sphere) -> "Hittable":
        return Hittable(A.RT_HITTABLE_SPHERE, sphere=s)

    @staticmethod
    def UnboundedSphere(s: Sphere) -> "Hittable":
        return Hittable(A.RT_HITTABLE_UNBOUNDED_SPHERE, sphere=s)

    @staticmethod
    def InfinitePlane(p: InfinitePlane) -> "Hittable":
        return Hittable(A.RT_HITTABLE_INFINITE_PLANE, plane=p)


# ---- Camera.fs ----------------------------------------------------------------------------------------------------
@dataclasses.dataclass(frozen=True)
class Camera:  # Camera.fs:3-28; `dataclasses.replace(camera, BounceDepth=50)` is F#'s `{ camera with BounceDepth = 50 }`
    abi: A.rt_camera = dataclasses.field(repr=False, compare=False)
    SamplesPerPixel: int = 1
    BounceDepth: int = 150

    @staticmethod
    def makeBasic(samplesPerPixel: int, focalLength: float, aspectRatio: float, origin: Point, viewDirection: UnitVector,
                  viewUp: Vector) -> "Camera":
        out = A.rt_camera()
        check(lib.rt_camera_make_basic(int(samplesPerPixel), float(focalLength), float(aspectRatio), (C.c_double * 3)(*origin),
                                       (C.c_double * 3)(*viewDirection), (C.c_double * 3)(*viewUp), C.byref(out)))
        return Camera(out, SamplesPerPixel=int(samplesPerPixel), BounceDepth=int(out.bounce_depth))

    def to_abi(self) -> A.rt_camera:
        c = A.rt_camera()
        C.memmove(C.byref(c), C.byref(self.abi), C.sizeof(A.rt_camera))
        c.samples_per_pixel = int(self.SamplesPerPixel)
        c.bounce_depth = int(self.BounceDepth)
        return c

    ViewportWidth = property(lambda self: self.abi.viewport_width)
    ViewportHeight = property(lambda self: self.abi.viewport_height)
    FocalLength = property(lambda self: self.abi.focal_length)


# ---- flatten hittables into the ABI arrays --------------------------------------------------------------------------
def flatten_hittables(objects: Sequence[Hittable]):
    """-> (rt_hittable array, rt_texture array, keepalive list)."""
    texs: List[A.rt_texture] = []
    keep: List[np.ndarray] = []

    def add_param(t: _ParamTex) -> int:
        r = A.rt_texture()
        r.kind = t.kind
        r.even = r.odd = -1
        r.map_radius = 1.0
        if t.kind == A.RT_TEXTURE_CHECKERED:
            e, o = add_param(t.even), add_param(t.odd)  # children first: indices smaller than the parent's
            r.even, r.odd, r.grid_size = e, o, t.grid
        elif t.kind == A.RT_TEXTURE_IMAGE:
            keep.append(t.image)
            r.height, r.width = int(t.image.shape[0]), int(t.image.shape[1])
            r.texels = t.image.ctypes.data
        elif t.kind == A.RT_TEXTURE_UV_RAMP:
            r.ramp_src[:] = t.ramp
        r.rgb[:] = t.pixel
        texs.append(r)
        return len(texs) - 1

    hs = (A.rt_hittable * max(1, len(objects)))()
    for i, h in enumerate(objects):
        o = hs[i]
        o.kind = h.kind
        o.texture = -1
        st = h.plane.Style if h.kind == A.RT_HITTABLE_INFINITE_PLANE else h.sphere.Style
        o.style = st.style
        o.albedo, o.fuzz, o.ior, o.prob = st.albedo, st.fuzz, st.ior, st.prob
        colour = st.colour
        if st.texture is not None:
            if st.texture.pixel is not None:
                colour = st.texture.pixel
            else:
                idx = add_param(st.texture.param)
                texs[idx].map_radius = st.texture.map_radius
                texs[idx].map_centre[:] = st.texture.map_centre
                o.texture = idx
        o.rgb[:] = colour
        if h.kind == A.RT_HITTABLE_INFINITE_PLANE:
            o.point[:] = h.plane.Point
            o.normal[:] = h.plane.Normal
        else:
            o.point[:] = h.sphere.Centre
            o.radius = h.sphere.Radius
    tex_arr = (A.rt_texture * max(1, len(texs)))(*texs) if texs else (A.rt_texture * 1)()
    return hs, len(objects), tex_arr, len(texs), keep


# ---- Domain.fs ----------------------------------------------------------------------------------------------------
class Image:  # Domain.fs:9-31: rows are produced lazily, on first use
    def __init__(self, rowCount: int, colCount: int, force: Callable[[], np.ndarray]):
        self.RowCount, self.ColCount = rowCount, colCount
        self._force, self._rows = force, None

    @staticmethod
    def rowCount(i: "Image") -> int:
        return i.RowCount

    @staticmethod
    def colCount(i: "Image") -> int:
        return i.ColCount

    @staticmethod
    def make(rowCount: int, colCount: int, pixels) -> "Image":
        arr = np.asarray(pixels, dtype=np.uint8).reshape(rowCount, colCount, 3)
        return Image(rowCount, colCount, lambda: arr)

    @staticmethod
    def render(i: "Image") -> np.ndarray:
        """Image.render (Domain.fs:23-24): forces the rows; [RowCount, ColCount, 3] uint8, row 0 = top."""
        if i._rows is None:
            i._rows = i._force()
        return i._rows


# ---- Scene.fs -----------------------------------------------------------------------------------------------------
class RenderResult(NamedTuple):
    accum: np.ndarray  # [n_rows, cols, 4] int32: PixelStats {Count; SumRed; SumGreen; SumBlue}
    rgb: np.ndarray    # [n_rows, cols, 3] uint8: PixelStats.mean
    stats: dict


class Scene:
    def __init__(self, handle: int, keep):
        self._h = C.c_void_p(handle)
        self._keep = keep
        self.last_stats: Optional[dict] = None

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            lib.rt_scene_destroy(h)

    @staticmethod
    def make(objects: Sequence[Hittable], walk_tree: Optional[str] = None) -> "Scene":  # Scene.fs:15-28
        """walk_tree: None = the process default (set_walk_tree), "sah" or "reference" = this scene's own choice (rt_scene_create_ex)."""
        hs, n, tex, ntex, keep = flatten_hittables(objects)
        out = C.c_void_p()
        if walk_tree is None:
            check(lib.rt_scene_create(hs, n, tex, ntex, C.byref(out)))
        else:
            opt = A.rt_scene_options({"sah": A.RT_WALK_TREE_SAH, "reference": A.RT_WALK_TREE_REFERENCE}[walk_tree])
            check(lib.rt_scene_create_ex(hs, n, tex, ntex, C.byref(opt), C.byref(out)))
        return Scene(out.value, keep)

    @property
    def handle(self) -> C.c_void_p:
        return self._h

    def info(self) -> dict:
        i = A.rt_scene_info()
        check(lib.rt_scene_get_info(self._h, C.byref(i)))
        return {n: getattr(i, n) for n, _ in i._fields_}

    def tree(self):
        n = self.info()["n_nodes"]
        skip = np.zeros(n, np.int32); prim = np.zeros(n, np.int32); boxes = np.zeros((n, 6), np.float64)
        check(lib.rt_scene_get_tree(self._h, _i32(skip), _i32(prim), _f64(boxes)))
        return skip, prim, boxes

    def walk_tree(self):
        """The tree the device image holds (rt_set_walk_tree): same arrays as tree()."""
        n = self.info()["walk_tree_nodes"]
        skip = np.zeros(n, np.int32); prim = np.zeros(n, np.int32); boxes = np.zeros((n, 6), np.float64)
        check(lib.rt_scene_get_walk_tree(self._h, _i32(skip), _i32(prim), _f64(boxes)))
        return skip, prim, boxes

    def filter_tree(self):
        """rt_scene_get_filter_tree: the timed kernel's single-precision records in the device image's (depth) order:
        boxes [n, 6] float32 (lo, hi per axis, rounded outward), links [n, 5] int32 (on hit, on miss, queue entry, shift, hittable)."""
        n = self.info()["walk_tree_nodes"]
        boxes = np.zeros((n, 6), np.float32); links = np.zeros((n, 5), np.int32)
        check(lib.rt_scene_get_filter_tree(self._h, boxes.ctypes.data_as(C.POINTER(C.c_float)), _i32(links)))
        return boxes, links

    def tune(self, maxWidthCoord: int, maxHeightCoord: int, camera: Camera, *, seed: int = 0, device: int = 0) -> dict:
        """rt_scene_tune: the walk tree rebuilt from the rays of a small probe render with this camera (same pixels, fewer box tests)."""
        info = A.rt_tune_info()
        check(lib.rt_scene_tune(self._h, C.byref(camera.to_abi()), maxWidthCoord, maxHeightCoord, seed, device, C.byref(info)))
        return {n: getattr(info, n) for n, _ in info._fields_ if n != "struct_size"}

    def tune_rays(self, rays: np.ndarray) -> dict:
        """rt_scene_tune_rays: the host half of tune() with the caller's own probe rays [n, 6] (origin, direction)."""
        r = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 6)
        info = A.rt_tune_info()
        check(lib.rt_scene_tune_rays(self._h, _f64(r), r.shape[0], C.byref(info)))
        return {n: getattr(info, n) for n, _ in info._fields_ if n != "struct_size"}

    def render_rows(self, maxWidthCoord: int, maxHeightCoord: int, camera: Camera, *, seed: int = 0, device: int = 0,
                    row_first: int = 0, row_stride: int = 1, n_rows: Optional[int] = None, counters: bool = False) -> RenderResult:
        """rt_render for the image rows row_first + i*row_stride (the whole frame by default)."""
        rows = 2 * maxHeightCoord + 1
        cols = 2 * maxWidthCoord + 1
        if n_rows is None:
            n_rows = max(0, (rows - row_first + row_stride - 1) // row_stride)
        accum = np.zeros((n_rows, cols, 4), np.int32)
        rgb = np.zeros((n_rows, cols, 3), np.uint8)
        st = A.rt_stats()
        cam = camera.to_abi()
        check(lib.rt_render(self._h, C.byref(cam), maxWidthCoord, maxHeightCoord, seed, device, row_first, row_stride, n_rows,
                            A.RT_RENDER_COUNTERS if counters else 0, _i32(accum), _u8(rgb), C.byref(st)))
        self.last_stats = st.as_dict()
        return RenderResult(accum, rgb, self.last_stats)

    def render_frame(self, maxWidthCoord: int, maxHeightCoord: int, camera: Camera, *, seed: int = 0, devices: Sequence[int] = (0,),
                     gather: int = A.RT_GATHER_AUTO, counters: bool = False, options: Optional[A.rt_render_options] = None) -> RenderResult:
        """rt_render_frame: the whole frame on several GPUs from this one process (rows interleaved over `devices`, one gather);
        stats is a list with one dict per device."""
        rows, cols = 2 * maxHeightCoord + 1, 2 * maxWidthCoord + 1
        accum = np.zeros((rows, cols, 4), np.int32)
        rgb = np.zeros((rows, cols, 3), np.uint8)
        devs = (C.c_int32 * len(devices))(*devices)
        st = (A.rt_stats * len(devices))()
        cam = camera.to_abi()
        check(lib.rt_render_frame(self._h, C.byref(cam), maxWidthCoord, maxHeightCoord, seed, devs, len(devices),
                                  A.RT_RENDER_COUNTERS if counters else 0, gather, C.byref(options) if options is not None else None,
                                  _i32(accum), _u8(rgb), st))
        return RenderResult(accum, rgb, [x.as_dict() for x in st])

    @staticmethod
    def render(progressIncrement: Callable[[float], None], log: Callable[[str], None], maxWidthCoord: int, maxHeightCoord: int,
               camera: Camera, s: "Scene", *, seed: int = 0, device: int = 0, row_block: Optional[int] = None) -> Tuple[float, Image]:
        """Scene.render (Scene.fs:196-236): returns (rows as progress units, lazy Image).  The work happens when the
        Image is forced, exactly as in the reference, and `progressIncrement 1.0` is called once per row (Scene.fs:232) -- after
        each block of `row_block` rows has come back (row_first/n_rows of the ABI), or after the whole frame when row_block is
        None (one launch: fastest; at 0.2 s per frame the bar just jumps)."""
        rowsIter = 2 * maxHeightCoord + 1
        colsIter = 2 * maxWidthCoord + 1

        def force() -> np.ndarray:
            block = rowsIter if not row_block else max(1, int(row_block))
            out = np.zeros((rowsIter, colsIter, 3), np.uint8)
            for first in range(0, rowsIter, block):
                n = min(block, rowsIter - first)
                out[first:first + n] = s.render_rows(maxWidthCoord, maxHeightCoord, camera, seed=seed, device=device, row_first=first,
                                                     row_stride=1, n_rows=n).rgb
                for _ in range(n):
                    progressIncrement(1.0)
            return out

        return float(rowsIter), Image(rowsIter, colsIter, force)


# ---- ImageOutput.fs -------------------------------------------------------------------------------------------------
class PixelOutput:
    @staticmethod
    def correct(b: int) -> int:  # ImageOutput.fs:11-18
        return int(lib.rt_gamma_correct(int(b)))


class ImageOutput:
    @staticmethod
    def formatPpm(gammaCorrect: bool, pixels: np.ndarray) -> bytes:
        px = np.ascontiguousarray(pixels, dtype=np.uint8)
        rows, cols = px.shape[0], px.shape[1]
        n = lib.rt_format_ppm(_u8(px), rows, cols, int(bool(gammaCorrect)), None, 0)
        if n < 0:
            check(int(-n))
        buf = C.create_string_buffer(int(n) + 1)
        lib.rt_format_ppm(_u8(px), rows, cols, int(bool(gammaCorrect)), buf, int(n) + 1)
        return buf.raw[: int(n)]

    @staticmethod
    def toPpm(progressIncrement: Callable[[float], None], image: "Image", path: str) -> str:
        """ImageOutput.toPpm = resume (ImageOutput.fs:131-161, 205): force the image and spill it as `<row>,<col>\\nRGB` records."""
        px = np.ascontiguousarray(Image.render(image), dtype=np.uint8)
        n = lib.rt_format_pixel_map(_u8(px), px.shape[0], px.shape[1], None, 0)
        if n < 0:
            check(int(-n))
        buf = C.create_string_buffer(int(n))
        lib.rt_format_pixel_map(_u8(px), px.shape[0], px.shape[1], buf, int(n))
        with open(path, "wb") as f:
            f.write(buf.raw[: int(n)])
        for _ in range(px.shape[0]):
            progressIncrement(1.0)
        return path

    @staticmethod
    def readPixelMap(path: str, numRows: int, numCols: int):
        """ImageOutput.readPixelMap (ImageOutput.fs:68-113) -> (pixels [rows, cols, 3] uint8, present [rows, cols] bool)."""
        data = open(path, "rb").read()
        rgb = np.zeros((numRows, numCols, 3), np.uint8)
        present = np.zeros((numRows, numCols), np.uint8)
        n = lib.rt_parse_pixel_map(data, len(data), numRows, numCols, _u8(rgb), _u8(present))
        if n < 0:
            check(int(-n))
        return rgb, present.astype(bool)

    @staticmethod
    def assertComplete(pixel_map) -> np.ndarray:
        """ImageOutput.assertComplete (ImageOutput.fs:199-200): ValueOption.get on every pixel."""
        rgb, present = pixel_map
        if not present.all():
            raise ValueError("ValueOption.get: the pixel map is incomplete")
        return rgb

    @staticmethod
    def writePpm(gammaCorrect: bool, incrementProgress: Callable[[float], None], pixels: np.ndarray, output: str) -> None:
        """ImageOutput.writePpm (ImageOutput.fs:163-197)."""
        px = np.ascontiguousarray(pixels, dtype=np.uint8)
        check(lib.rt_write_ppm(str(output).encode(), _u8(px), px.shape[0], px.shape[1], int(bool(gammaCorrect))))
        for _ in range(px.shape[0] * px.shape[1]):
            incrementProgress(1.0)


def _f64(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _i32(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _u8(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def _u32(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def _u64(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_uint64))
