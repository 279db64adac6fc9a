"""Loads librtfs_amd.so (the HIP product library) and declares the C ABI of include/rtfs_amd.h.

There is no fallback: if the library is missing the import fails loudly, and compute entry points return
RT_ERR_NO_DEVICE (raised as RtError) when no GPU is visible.
"""
import ctypes as C
import os

from . import _abi as A

_HERE = os.path.dirname(os.path.abspath(__file__))
# RTFS_LIB: another build of the same library (kernel A/B experiments, scripts/ab.sh); the default is the in-tree product build
LIB_PATH = os.environ.get("RTFS_LIB") or os.path.join(_HERE, "librtfs_amd.so")


class RtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"rtfs_amd error {code}: {message}")
        self.code = code
        self.message = message


def _preload_hip_runtime():
    """A process must hold ONE HIP/HSA runtime.  PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME libamdhip64.so.7,
    RPATH $ORIGIN); if librtfs_amd.so pulled in /opt/rocm's copy first, a later `import torch` would start a second runtime
    that finds no GPU.  So when a torch wheel is present, its runtime is loaded first (torch itself is NOT imported) and
    librtfs_amd.so's NEEDED libamdhip64.so.7 binds to it; without torch the system runtime is used."""
    import importlib.util

    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    bundled = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(bundled):
        C.CDLL(bundled, mode=C.RTLD_GLOBAL)


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C ray-tracing-fsharp_amd/csrc` (hipcc, gfx950). There is no CPU fallback."
        )
    _preload_hip_runtime()
    return C.CDLL(LIB_PATH)


lib = _load()

_P = C.POINTER
_dp, _u8p, _i32p, _u32p, _u64p = _P(C.c_double), _P(C.c_uint8), _P(C.c_int32), _P(C.c_uint32), _P(C.c_uint64)

# name -> (restype, argtypes); every symbol include/rtfs_amd.h declares
SIGNATURES = {
    "rt_abi_version": (C.c_int, []),
    "rt_abi_sizeof": (C.c_size_t, [C.c_int]),
    "rt_abi_offsetof": (C.c_size_t, [C.c_int, C.c_int]),
    "rt_last_error": (C.c_char_p, []),
    "rt_device_count": (C.c_int, []),
    "rt_set_launch_config": (C.c_int, [C.c_int32, C.c_int32, C.c_int32]),
    "rt_set_schedule": (C.c_int, [C.c_int32, C.c_int32]),
    "rt_set_passes": (C.c_int, [C.c_int32]),
    "rt_set_walk_tree": (C.c_int, [C.c_int32]),
    "rt_set_park": (C.c_int, [C.c_int32]),
    "rt_last_stage_stats": (C.c_int, [_u64p]),  # out[16]
    "rt_camera_make_basic": (C.c_int, [C.c_int32, C.c_double, C.c_double, _dp, _dp, _dp, _P(A.rt_camera)]),
    "rt_scene_create": (C.c_int, [_P(A.rt_hittable), C.c_size_t, _P(A.rt_texture), C.c_size_t, _P(C.c_void_p)]),
    "rt_scene_create_ex": (C.c_int, [_P(A.rt_hittable), C.c_size_t, _P(A.rt_texture), C.c_size_t, _P(A.rt_scene_options), _P(C.c_void_p)]),
    "rt_scene_destroy": (None, [C.c_void_p]),
    "rt_scene_get_info": (C.c_int, [C.c_void_p, _P(A.rt_scene_info)]),
    "rt_scene_get_tree": (C.c_int, [C.c_void_p, _i32p, _i32p, _dp]),
    "rt_scene_get_walk_tree": (C.c_int, [C.c_void_p, _i32p, _i32p, _dp]),
    "rt_scene_get_filter_tree": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), _i32p]),
    "rt_scene_tune_rays": (C.c_int, [C.c_void_p, _dp, C.c_size_t, _P(A.rt_tune_info)]),
    "rt_scene_tune": (C.c_int, [C.c_void_p, _P(A.rt_camera), C.c_int32, C.c_int32, C.c_uint64, C.c_int32, _P(A.rt_tune_info)]),
    "rt_render": (C.c_int, [C.c_void_p, _P(A.rt_camera), C.c_int32, C.c_int32, C.c_uint64, C.c_int32, C.c_int32, C.c_int32,
                            C.c_int32, C.c_uint32, _i32p, _u8p, _P(A.rt_stats)]),
    "rt_render_device": (C.c_int, [C.c_void_p, _P(A.rt_camera), C.c_int32, C.c_int32, C.c_uint64, C.c_int32, C.c_int32,
                                   C.c_int32, C.c_int32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, _P(A.rt_stats)]),
    "rt_render_device_ex": (C.c_int, [C.c_void_p, _P(A.rt_camera), C.c_int32, C.c_int32, C.c_uint64, C.c_int32, C.c_int32,
                                      C.c_int32, C.c_int32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, _P(A.rt_render_options),
                                      _P(A.rt_stats)]),
    "rt_render_frame": (C.c_int, [C.c_void_p, _P(A.rt_camera), C.c_int32, C.c_int32, C.c_uint64, _i32p, C.c_int32, C.c_uint32, C.c_int32,
                                  _P(A.rt_render_options), _i32p, _u8p, _P(A.rt_stats)]),
    "rt_gamma_correct": (C.c_uint8, [C.c_uint8]),
    "rt_write_ppm": (C.c_int, [C.c_char_p, _u8p, C.c_int32, C.c_int32, C.c_int32]),
    "rt_format_ppm": (C.c_int64, [_u8p, C.c_int32, C.c_int32, C.c_int32, C.c_char_p, C.c_size_t]),
    "rt_format_pixel_map": (C.c_int64, [_u8p, C.c_int32, C.c_int32, C.c_char_p, C.c_size_t]),
    "rt_parse_pixel_map": (C.c_int64, [C.c_char_p, C.c_size_t, C.c_int32, C.c_int32, _u8p, _u8p]),
    "rt_dev_float_producer": (C.c_int, [C.c_int32, _u32p, C.c_int32, _dp]),
    "rt_dev_stream_state": (C.c_int, [C.c_int32, C.c_uint64, C.c_int32, _u64p, _u32p, _u32p]),
    "rt_dev_bbox_hits": (C.c_int, [C.c_int32, C.c_int32, _dp, _dp, _i32p]),
    "rt_dev_bbox_filter": (C.c_int, [C.c_int32, C.c_int32, _dp, _dp, C.c_double, _i32p]),
    "rt_dev_pixel_candidates": (C.c_int, [C.c_int32, C.c_void_p, _P(A.rt_camera), C.c_int32, C.c_int32, C.c_int32, _i32p, _i32p]),
    "rt_dev_sphere_first_intersection": (C.c_int, [C.c_int32, C.c_int32, _dp, _dp, _dp]),
    "rt_dev_plane_intersection": (C.c_int, [C.c_int32, C.c_int32, _dp, _dp, _dp]),
    "rt_dev_pixel_combine": (C.c_int, [C.c_int32, C.c_int32, _u8p, _u8p, _u8p]),
    "rt_dev_pixel_darken": (C.c_int, [C.c_int32, C.c_int32, _u8p, _dp, _u8p]),
    "rt_dev_reflection": (C.c_int, [C.c_int32, C.c_void_p, C.c_int32, _i32p, _dp, _u8p, _dp, _u32p, _i32p, _u8p, _dp]),
    "rt_dev_hit_object": (C.c_int, [C.c_int32, C.c_void_p, C.c_int32, _dp, _i32p, _dp, _u32p]),
    "rt_dev_hit_object_lds": (C.c_int, [C.c_int32, C.c_void_p, C.c_int32, _dp, _i32p, _dp]),
    "rt_dev_trace_ray": (C.c_int, [C.c_int32, C.c_void_p, C.c_int32, C.c_int32, _dp, _u32p, _u8p]),
    "rt_dev_texture_colour_at": (C.c_int, [C.c_int32, C.c_void_p, C.c_int32, C.c_int32, _dp, _dp, _u8p]),
    "rt_dev_arith": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _dp, _dp, _dp]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)  # AttributeError here = the library does not export a declared symbol
    _fn.restype = _res
    _fn.argtypes = _args


def check(code):
    if code != A.RT_OK:
        raise RtError(code, (lib.rt_last_error() or b"").decode("utf-8", "replace"))


for _i, _t in enumerate(A.ABI_STRUCTS):  # sizes and every field offset of the ctypes mirrors against the library as compiled
    if lib.rt_abi_sizeof(_i) != C.sizeof(_t):
        raise ImportError(f"ABI mismatch for {_t.__name__}: library {lib.rt_abi_sizeof(_i)} vs ctypes {C.sizeof(_t)}")
    for _k, (_fname, *_rest) in enumerate(_t._fields_):
        if lib.rt_abi_offsetof(_i, _k) != getattr(_t, _fname).offset:
            raise ImportError(f"ABI mismatch for {_t.__name__}.{_fname}: library offset {lib.rt_abi_offsetof(_i, _k)} vs ctypes {getattr(_t, _fname).offset}")
    if lib.rt_abi_offsetof(_i, len(_t._fields_)) != C.c_size_t(-1).value:
        raise ImportError(f"ABI mismatch for {_t.__name__}: the library has more fields than the ctypes mirror")
if lib.rt_abi_version() != A.RT_ABI_VERSION:
    raise ImportError("ABI version mismatch between librtfs_amd.so and _abi.py")
