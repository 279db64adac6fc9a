"""numpy wrappers over the rt_dev_* device unit hooks (include/rtfs_amd.h): each runs ONE device function of the
path on the GPU so the reference's unit tests can be replayed against the code the render kernel inlines."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import check, lib
from .raytracing import Scene, _f64, _i32, _u8, _u32, _u64


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def float_producer(state, n, device=0):
    out = np.zeros(n, np.float64)
    check(lib.rt_dev_float_producer(device, (C.c_uint32 * 4)(*[int(s) for s in state]), n, _f64(out)))
    return out


def stream_state(seed, pixel, sample, device=0):
    pixel, sample = _c(pixel, np.uint64), _c(sample, np.uint32)
    out = np.zeros((len(pixel), 4), np.uint32)
    check(lib.rt_dev_stream_state(device, int(seed), len(pixel), _u64(pixel), _u32(sample), _u32(out)))
    return out


def bbox_hits(rays, boxes, device=0):
    rays, boxes = _c(rays, np.float64).reshape(-1, 6), _c(boxes, np.float64).reshape(-1, 6)
    out = np.zeros(len(rays), np.int32)
    check(lib.rt_dev_bbox_hits(device, len(rays), _f64(rays), _f64(boxes), _i32(out)))
    return out


def bbox_filter(rays, boxes, bmax=0.0, device=0):
    """bit 0: BoundingBox.hits, exactly; bit 1: the timed node loop's single-precision filter (its own instructions); bit 2: the same
    filter as compiled C++.  The filter must say 'hit' wherever the exact test does."""
    rays, boxes = _c(rays, np.float64).reshape(-1, 6), _c(boxes, np.float64).reshape(-1, 6)
    out = np.zeros(len(rays), np.int32)
    check(lib.rt_dev_bbox_filter(device, len(rays), _f64(rays), _f64(boxes), float(bmax), _i32(out)))
    return out


def sphere_first_intersection(rays, spheres, device=0):
    rays, spheres = _c(rays, np.float64).reshape(-1, 6), _c(spheres, np.float64).reshape(-1, 4)
    out = np.zeros(len(rays), np.float64)
    check(lib.rt_dev_sphere_first_intersection(device, len(rays), _f64(rays), _f64(spheres), _f64(out)))
    return out


def plane_intersection(rays, planes, device=0):
    rays, planes = _c(rays, np.float64).reshape(-1, 6), _c(planes, np.float64).reshape(-1, 6)
    out = np.zeros(len(rays), np.float64)
    check(lib.rt_dev_plane_intersection(device, len(rays), _f64(rays), _f64(planes), _f64(out)))
    return out


def pixel_combine(a, b, device=0):
    a, b = _c(a, np.uint8).reshape(-1, 3), _c(b, np.uint8).reshape(-1, 3)
    out = np.zeros_like(a)
    check(lib.rt_dev_pixel_combine(device, len(a), _u8(a), _u8(b), _u8(out)))
    return out


def pixel_darken(p, albedo, device=0):
    p, albedo = _c(p, np.uint8).reshape(-1, 3), _c(albedo, np.float64)
    out = np.zeros_like(p)
    check(lib.rt_dev_pixel_darken(device, len(p), _u8(p), _f64(albedo), _u8(out)))
    return out


def arith(op, a, b=None, device=0):
    a = _c(a, np.float64)
    bb = _c(b, np.float64) if b is not None else None
    out = np.zeros_like(a)
    check(lib.rt_dev_arith(device, op, len(a), _f64(a), _f64(bb) if bb is not None else None, _f64(out)))
    return out


def reflection(scene: Scene, index, ray_in, colour_in, strike, rng_state, device=0):
    index = _c(index, np.int32)
    ray_in, strike = _c(ray_in, np.float64).reshape(-1, 6), _c(strike, np.float64).reshape(-1, 3)
    colour_in = _c(colour_in, np.uint8).reshape(-1, 3)
    rng = _c(rng_state, np.uint32).reshape(-1, 4).copy()
    n = len(index)
    absorbed, col, ray = np.zeros(n, np.int32), np.zeros((n, 3), np.uint8), np.zeros((n, 6), np.float64)
    check(lib.rt_dev_reflection(device, scene.handle, n, _i32(index), _f64(ray_in), _u8(colour_in), _f64(strike), _u32(rng),
                                _i32(absorbed), _u8(col), _f64(ray)))
    return absorbed, col, ray, rng


def hit_object(scene: Scene, rays, device=0):
    rays = _c(rays, np.float64).reshape(-1, 6)
    n = len(rays)
    hit, strike, cnt = np.zeros(n, np.int32), np.zeros((n, 3), np.float64), np.zeros((n, 2), np.uint32)
    check(lib.rt_dev_hit_object(device, scene.handle, n, _f64(rays), _i32(hit), _f64(strike), _u32(cnt)))
    return hit, strike, cnt


def hit_object_lds(scene: Scene, rays, device=0):
    """Scene.hitObject through the timed kernel variant's route: scene in LDS, hand-written node loop."""
    rays = _c(rays, np.float64).reshape(-1, 6)
    n = len(rays)
    hit, strike = np.zeros(n, np.int32), np.zeros((n, 3), np.float64)
    check(lib.rt_dev_hit_object_lds(device, scene.handle, n, _f64(rays), _i32(hit), _f64(strike)))
    return hit, strike


def pixel_candidates(scene: Scene, camera, max_w, max_h, row_col, device=0):
    """The Leaves the camera rays of each pixel can reach, as the timed kernel finds them once per pixel: [n, 4] hittable indices,
    -1 padded; a row starting with -2 means "these rays walk the tree"."""
    import ctypes as C
    rc = _c(row_col, np.int32).reshape(-1, 2)
    out = np.zeros((len(rc), 4), np.int32)
    cam = camera.to_abi()
    check(lib.rt_dev_pixel_candidates(device, scene.handle, C.byref(cam), int(max_w), int(max_h), len(rc), _i32(rc), _i32(out)))
    return out


def trace_ray(scene: Scene, bounce_depth, rays, rng_state, device=0):
    rays = _c(rays, np.float64).reshape(-1, 6)
    rng = _c(rng_state, np.uint32).reshape(-1, 4).copy()
    col = np.zeros((len(rays), 3), np.uint8)
    check(lib.rt_dev_trace_ray(device, scene.handle, int(bounce_depth), len(rays), _f64(rays), _u32(rng), _u8(col)))
    return col, rng


def texture_colour_at(scene: Scene, texture, points, device=0):
    points = _c(points, np.float64).reshape(-1, 3)
    uv, col = np.zeros((len(points), 2), np.float64), np.zeros((len(points), 3), np.uint8)
    check(lib.rt_dev_texture_colour_at(device, scene.handle, int(texture), len(points), _f64(points), _f64(uv), _u8(col)))
    return uv, col
