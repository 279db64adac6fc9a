"""ctypes loader for the CPU oracle (oracle/oracle.cpp).  TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg import this package; the product package never does.

The oracle takes the same POD structs as the product ABI (include/rtfs_amd.h), so tests hand both sides identical
bytes; the struct mirrors are borrowed from ray_tracing_fsharp_amd._abi (an interface description, not product logic).
"""
import ctypes as C
import os
import subprocess

import numpy as np

import ray_tracing_fsharp_amd._abi as A
from ray_tracing_fsharp_amd.raytracing import flatten_hittables

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "oracle.cpp")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return LIB_PATH


def _load():
    if not os.path.exists(LIB_PATH):
        build()
    return C.CDLL(LIB_PATH)


lib = _load()
_P = C.POINTER
_dp, _u8p, _i32p, _u32p, _u64p = _P(C.c_double), _P(C.c_uint8), _P(C.c_int32), _P(C.c_uint32), _P(C.c_uint64)
for _n, _r, _a in [
    ("orc_last_error", C.c_char_p, []),
    ("orc_camera_make_basic", C.c_int, [C.c_int32, C.c_double, C.c_double, _dp, _dp, _dp, _P(A.rt_camera)]),
    ("orc_scene_create", C.c_int, [_P(A.rt_hittable), C.c_size_t, _P(A.rt_texture), C.c_size_t, _P(C.c_void_p)]),
    ("orc_scene_destroy", None, [C.c_void_p]),
    ("orc_scene_get_tree", C.c_int, [C.c_void_p, _i32p, _i32p, _i32p, _i32p, _dp]),
    ("orc_render", C.c_int, [C.c_void_p, _P(A.rt_camera), C.c_int32, C.c_int32, C.c_uint64, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                             _i32p, _u8p, _P(A.rt_stats)]),
    ("orc_set_brute_force", None, [C.c_int]),
    ("orc_gamma_correct", C.c_uint8, [C.c_uint8]),
    ("orc_format_ppm", C.c_int64, [_u8p, C.c_int32, C.c_int32, C.c_int32, C.c_char_p, C.c_size_t]),
    ("orc_format_pixel_map", C.c_int64, [_u8p, C.c_int32, C.c_int32, C.c_char_p, C.c_size_t]),
    ("orc_parse_pixel_map", C.c_int64, [C.c_char_p, C.c_size_t, C.c_int32, C.c_int32, _u8p, _u8p]),
    ("orc_float_producer", C.c_int, [_u32p, C.c_int32, _dp]),
    ("orc_stream_state", C.c_int, [C.c_uint64, C.c_int32, _u64p, _u32p, _u32p]),
    ("orc_bbox_hits", C.c_int, [C.c_int32, _dp, _dp, _i32p]),
    ("orc_sphere_first_intersection", C.c_int, [C.c_int32, _dp, _dp, _dp]),
    ("orc_plane_intersection", C.c_int, [C.c_int32, _dp, _dp, _dp]),
    ("orc_pixel_combine", C.c_int, [C.c_int32, _u8p, _u8p, _u8p]),
    ("orc_pixel_darken", C.c_int, [C.c_int32, _u8p, _dp, _u8p]),
    ("orc_reflection", C.c_int, [C.c_void_p, C.c_int32, _i32p, _dp, _u8p, _dp, _u32p, _i32p, _u8p, _dp]),
    ("orc_hit_object", C.c_int, [C.c_void_p, C.c_int32, _dp, _i32p, _dp, _u32p]),
    ("orc_trace_ray", C.c_int, [C.c_void_p, C.c_int32, C.c_int32, _dp, _u32p, _u8p]),
    ("orc_texture_colour_at", C.c_int, [C.c_void_p, C.c_int32, C.c_int32, _dp, _dp, _u8p]),
    ("orc_arith", C.c_int, [C.c_int32, C.c_int32, _dp, _dp, _dp]),
    ("orc_plane_map", C.c_int, [C.c_double, _dp, C.c_double, C.c_double, _dp]),
    ("orc_plane_map_inverse", C.c_int, [C.c_double, _dp, _dp, _dp]),
    ("orc_sphere_lies_on", C.c_int, [_dp, _dp, C.c_double]),
    ("orc_ray_make", C.c_int, [_dp, _dp, _dp]),
    ("orc_ray_walk_along", C.c_int, [_dp, C.c_double, _dp]),
    ("orc_ray_lies_on", C.c_int, [_dp, _dp]),
    ("orc_plane_orthonormal_basis", C.c_int, [_dp, _dp, _dp, _dp, _dp, _dp]),
    ("orc_hardware_threads", C.c_int, []),
]:
    _f = getattr(lib, _n)
    _f.restype, _f.argtypes = _r, _a


class OracleError(RuntimeError):
    pass


def _check(rc):
    if rc != 0:
        raise OracleError((lib.orc_last_error() or b"").decode())


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def _f64(a):
    return a.ctypes.data_as(_dp)


def _i32(a):
    return a.ctypes.data_as(_i32p)


def _u8(a):
    return a.ctypes.data_as(_u8p)


def _u32(a):
    return a.ctypes.data_as(_u32p)


def _u64(a):
    return a.ctypes.data_as(_u64p)


def _d3(v):
    return (C.c_double * 3)(*[float(x) for x in v])


def camera_make_basic(spp, focal, aspect, origin, view_dir, view_up) -> A.rt_camera:
    out = A.rt_camera()
    _check(lib.orc_camera_make_basic(spp, focal, aspect, _d3(origin), _d3(view_dir), _d3(view_up), C.byref(out)))
    return out


class OracleScene:
    """Scene.make on the oracle side, from the same host-mirror objects (or raw ABI arrays) the product takes."""

    def __init__(self, objects):
        hs, n, tex, ntex, keep = flatten_hittables(objects)
        self._keep = keep
        h = C.c_void_p()
        _check(lib.orc_scene_create(hs, n, tex, ntex, C.byref(h)))
        self._h = h

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            lib.orc_scene_destroy(h)

    def tree(self):
        n, d = C.c_int32(), C.c_int32()
        _check(lib.orc_scene_get_tree(self._h, C.byref(n), C.byref(d), None, None, None))
        skip, prim, boxes = np.zeros(n.value, np.int32), np.zeros(n.value, np.int32), np.zeros((n.value, 6), np.float64)
        _check(lib.orc_scene_get_tree(self._h, None, None, _i32(skip), _i32(prim), _f64(boxes)))
        return skip, prim, boxes, d.value

    def render_rows(self, max_w, max_h, camera_abi: A.rt_camera, seed=0, row_first=0, row_stride=1, n_rows=None, threads=1):
        rows, cols = 2 * max_h + 1, 2 * max_w + 1
        if n_rows is None:
            n_rows = max(0, (rows - row_first + row_stride - 1) // row_stride)
        accum, rgb = np.zeros((n_rows, cols, 4), np.int32), np.zeros((n_rows, cols, 3), np.uint8)
        st = A.rt_stats()
        _check(lib.orc_render(self._h, C.byref(camera_abi), max_w, max_h, seed, row_first, row_stride, n_rows, threads, _i32(accum), _u8(rgb),
                              C.byref(st)))
        return accum, rgb, st.as_dict()

    def reflection(self, index, ray_in, colour_in, strike, rng_state):
        index = _c(index, np.int32)
        ray_in, strike = _c(ray_in, np.float64).reshape(-1, 6), _c(strike, np.float64).reshape(-1, 3)
        colour_in = _c(colour_in, np.uint8).reshape(-1, 3)
        rng = _c(rng_state, np.uint32).reshape(-1, 4).copy()
        n = len(index)
        absorbed, col, ray = np.zeros(n, np.int32), np.zeros((n, 3), np.uint8), np.zeros((n, 6), np.float64)
        _check(lib.orc_reflection(self._h, n, _i32(index), _f64(ray_in), _u8(colour_in), _f64(strike), _u32(rng), _i32(absorbed), _u8(col), _f64(ray)))
        return absorbed, col, ray, rng

    def hit_object(self, rays):
        rays = _c(rays, np.float64).reshape(-1, 6)
        n = len(rays)
        hit, strike, cnt = np.zeros(n, np.int32), np.zeros((n, 3), np.float64), np.zeros((n, 2), np.uint32)
        _check(lib.orc_hit_object(self._h, n, _f64(rays), _i32(hit), _f64(strike), _u32(cnt)))
        return hit, strike, cnt

    def trace_ray(self, bounce_depth, rays, rng_state):
        rays = _c(rays, np.float64).reshape(-1, 6)
        rng = _c(rng_state, np.uint32).reshape(-1, 4).copy()
        col = np.zeros((len(rays), 3), np.uint8)
        _check(lib.orc_trace_ray(self._h, int(bounce_depth), len(rays), _f64(rays), _u32(rng), _u8(col)))
        return col, rng

    def texture_colour_at(self, texture, points):
        points = _c(points, np.float64).reshape(-1, 3)
        uv, col = np.zeros((len(points), 2), np.float64), np.zeros((len(points), 3), np.uint8)
        _check(lib.orc_texture_colour_at(self._h, int(texture), len(points), _f64(points), _f64(uv), _u8(col)))
        return uv, col


def float_producer(state, n):
    out = np.zeros(n, np.float64)
    _check(lib.orc_float_producer((C.c_uint32 * 4)(*[int(s) for s in state]), n, _f64(out)))
    return out


def stream_state(seed, pixel, sample):
    pixel, sample = _c(pixel, np.uint64), _c(sample, np.uint32)
    out = np.zeros((len(pixel), 4), np.uint32)
    _check(lib.orc_stream_state(int(seed), len(pixel), _u64(pixel), _u32(sample), _u32(out)))
    return out


def bbox_hits(rays, boxes):
    rays, boxes = _c(rays, np.float64).reshape(-1, 6), _c(boxes, np.float64).reshape(-1, 6)
    out = np.zeros(len(rays), np.int32)
    _check(lib.orc_bbox_hits(len(rays), _f64(rays), _f64(boxes), _i32(out)))
    return out


def sphere_first_intersection(rays, spheres):
    rays, spheres = _c(rays, np.float64).reshape(-1, 6), _c(spheres, np.float64).reshape(-1, 4)
    out = np.zeros(len(rays), np.float64)
    _check(lib.orc_sphere_first_intersection(len(rays), _f64(rays), _f64(spheres), _f64(out)))
    return out


def plane_intersection(rays, planes):
    rays, planes = _c(rays, np.float64).reshape(-1, 6), _c(planes, np.float64).reshape(-1, 6)
    out = np.zeros(len(rays), np.float64)
    _check(lib.orc_plane_intersection(len(rays), _f64(rays), _f64(planes), _f64(out)))
    return out


def pixel_combine(a, b):
    a, b = _c(a, np.uint8).reshape(-1, 3), _c(b, np.uint8).reshape(-1, 3)
    out = np.zeros_like(a)
    _check(lib.orc_pixel_combine(len(a), _u8(a), _u8(b), _u8(out)))
    return out


def pixel_darken(p, albedo):
    p, albedo = _c(p, np.uint8).reshape(-1, 3), _c(albedo, np.float64)
    out = np.zeros_like(p)
    _check(lib.orc_pixel_darken(len(p), _u8(p), _f64(albedo), _u8(out)))
    return out


def arith(op, a, b=None):
    a = _c(a, np.float64)
    bb = _c(b, np.float64) if b is not None else None
    out = np.zeros_like(a)
    _check(lib.orc_arith(op, len(a), _f64(a), _f64(bb) if bb is not None else None, _f64(out)))
    return out


def set_brute_force(on: bool) -> None:
    """Cross-check mode: hitObject tests every bounded sphere without any box (not the reference's algorithm)."""
    lib.orc_set_brute_force(int(bool(on)))


def gamma_correct(b):
    return int(lib.orc_gamma_correct(int(b)))


def format_ppm(pixels, gamma=False) -> bytes:
    px = _c(pixels, np.uint8)
    n = lib.orc_format_ppm(_u8(px), px.shape[0], px.shape[1], int(gamma), None, 0)
    buf = C.create_string_buffer(int(n) + 1)
    lib.orc_format_ppm(_u8(px), px.shape[0], px.shape[1], int(gamma), buf, int(n) + 1)
    return buf.raw[: int(n)]


def format_pixel_map(pixels) -> bytes:
    px = _c(pixels, np.uint8)
    n = lib.orc_format_pixel_map(_u8(px), px.shape[0], px.shape[1], None, 0)
    buf = C.create_string_buffer(int(n))
    lib.orc_format_pixel_map(_u8(px), px.shape[0], px.shape[1], buf, int(n))
    return buf.raw[: int(n)]


def parse_pixel_map(data: bytes, rows, cols):
    rgb, present = np.zeros((rows, cols, 3), np.uint8), np.zeros((rows, cols), np.uint8)
    n = lib.orc_parse_pixel_map(data, len(data), rows, cols, _u8(rgb), _u8(present))
    return int(n), rgb, present.astype(bool)


def plane_map(radius, centre, phi, theta):
    out = (C.c_double * 3)()
    lib.orc_plane_map(radius, _d3(centre), phi, theta, out)
    return tuple(out)


def plane_map_inverse(radius, centre, p):
    out = (C.c_double * 2)()
    lib.orc_plane_map_inverse(radius, _d3(centre), _d3(p), out)
    return tuple(out)


def sphere_lies_on(point, centre, radius) -> bool:
    return bool(lib.orc_sphere_lies_on(_d3(point), _d3(centre), radius))


def ray_make(origin, vec):
    out = (C.c_double * 6)()
    ok = lib.orc_ray_make(_d3(origin), _d3(vec), out)
    return tuple(out) if ok else None


def ray_walk_along(ray, magnitude):
    out = (C.c_double * 3)()
    lib.orc_ray_walk_along((C.c_double * 6)(*ray), magnitude, out)
    return tuple(out)


def ray_lies_on(point, ray) -> bool:
    return bool(lib.orc_ray_lies_on(_d3(point), (C.c_double * 6)(*ray)))


def plane_orthonormal_basis(origin, v1, v2, view_up):
    x, y = (C.c_double * 3)(), (C.c_double * 3)()
    ok = lib.orc_plane_orthonormal_basis(_d3(origin), _d3(v1), _d3(v2), _d3(view_up), x, y)
    return (tuple(x), tuple(y)) if ok else None


def hardware_threads() -> int:
    return int(lib.orc_hardware_threads())
