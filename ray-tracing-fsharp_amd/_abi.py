"""ctypes mirror of include/rtfs_amd.h (struct layouts, enums, status codes).  Keep in lock-step with the header;
tests/test_abi.py checks sizes against the values compiled into the library (rt_abi_sizeof)."""
import ctypes as C

RT_ABI_VERSION = 6

RT_OK, RT_ERR_INVALID_ARGUMENT, RT_ERR_NO_DEVICE, RT_ERR_HIP, RT_ERR_UNSUPPORTED, RT_ERR_IO = range(6)

# rt_hittable_kind
RT_HITTABLE_SPHERE, RT_HITTABLE_UNBOUNDED_SPHERE, RT_HITTABLE_INFINITE_PLANE = range(3)
# rt_sphere_style (Sphere.fs:10-37)
(RT_SPHERE_LIGHT_SOURCE, RT_SPHERE_LIGHT_SOURCE_CAP, RT_SPHERE_PURE_REFLECTION, RT_SPHERE_FUZZED_REFLECTION,
 RT_SPHERE_LAMBERT_REFLECTION, RT_SPHERE_DIELECTRIC, RT_SPHERE_GLASS) = range(7)
# rt_plane_style (InfinitePlane.fs:3-13)
RT_PLANE_LIGHT_SOURCE, RT_PLANE_PURE_REFLECTION, RT_PLANE_LAMBERT_REFLECTION, RT_PLANE_FUZZED_REFLECTION = range(4)
# rt_texture_kind
RT_TEXTURE_COLOUR, RT_TEXTURE_CHECKERED, RT_TEXTURE_IMAGE, RT_TEXTURE_UV_RAMP = range(4)
RT_RAMP_CONST, RT_RAMP_U, RT_RAMP_V = range(3)
RT_WALK_TREE_SAH, RT_WALK_TREE_REFERENCE, RT_WALK_TREE_TUNED = range(3)

RT_RENDER_COUNTERS = 1
RT_GATHER_AUTO, RT_GATHER_RCCL, RT_GATHER_PEER, RT_GATHER_HOST = range(4)

D3 = C.c_double * 3


class rt_hittable(C.Structure):
    _fields_ = [
        ("kind", C.c_uint32), ("style", C.c_uint32),
        ("point", D3), ("normal", D3),
        ("radius", C.c_double), ("albedo", C.c_double), ("fuzz", C.c_double), ("ior", C.c_double), ("prob", C.c_double),
        ("rgb", C.c_uint8 * 3), ("reserved", C.c_uint8),
        ("texture", C.c_int32),
    ]


class rt_texture(C.Structure):
    _fields_ = [
        ("kind", C.c_uint32),
        ("rgb", C.c_uint8 * 3), ("ramp_src", C.c_uint8 * 3), ("reserved", C.c_uint8 * 2),
        ("even", C.c_int32), ("odd", C.c_int32),
        ("grid_size", C.c_double),
        ("width", C.c_int32), ("height", C.c_int32),
        ("texels", C.c_void_p),
        ("map_centre", D3), ("map_radius", C.c_double),
    ]


class rt_camera(C.Structure):
    _fields_ = [
        ("view_origin", D3), ("view_dir", D3),
        ("xaxis_origin", D3), ("xaxis_dir", D3),
        ("yaxis_origin", D3), ("yaxis_dir", D3),
        ("viewport_width", C.c_double), ("viewport_height", C.c_double), ("focal_length", C.c_double),
        ("samples_per_pixel", C.c_int32), ("bounce_depth", C.c_int32),
    ]


class rt_scene_info(C.Structure):
    _fields_ = [
        ("n_bounded", C.c_int32), ("n_unbounded", C.c_int32), ("n_nodes", C.c_int32), ("tree_depth", C.c_int32),
        ("n_textures", C.c_int32), ("lds_resident", C.c_int32), ("walk_tree", C.c_int32), ("walk_tree_depth", C.c_int32),
        ("scene_bytes", C.c_int64), ("texel_bytes", C.c_int64), ("walk_tree_nodes", C.c_int32), ("leaf_box_implied", C.c_int32),
    ]


class rt_stats(C.Structure):
    _fields_ = [
        ("rays", C.c_uint64), ("aabb_tests", C.c_uint64), ("prim_tests", C.c_uint64), ("reflections", C.c_uint64),
        ("samples", C.c_uint64), ("pixels", C.c_uint64), ("pixels_early", C.c_uint64),
        ("kernel_ms", C.c_double), ("total_ms", C.c_double),
    ]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class rt_render_options(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("block_threads", C.c_int32), ("chunk_pixels", C.c_int32), ("blocks_per_cu", C.c_int32),
        ("yield_lanes", C.c_int32), ("refill_lanes", C.c_int32), ("passes", C.c_int32), ("park_lanes", C.c_int32),
    ]

    def __init__(self, **kw):
        super().__init__(struct_size=C.sizeof(rt_render_options), **kw)


class rt_scene_options(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("walk_tree", C.c_int32)]

    def __init__(self, walk_tree=-1):
        super().__init__(struct_size=C.sizeof(rt_scene_options), walk_tree=walk_tree)


class rt_tune_info(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("tuned", C.c_int32), ("probe_rows", C.c_int32), ("probe_rays", C.c_int32),
        ("nodes_before", C.c_int32), ("nodes_after", C.c_int32), ("box_tests_before", C.c_double), ("box_tests_after", C.c_double),
        ("probe_ms", C.c_double), ("build_ms", C.c_double),
    ]

    def __init__(self):
        super().__init__(struct_size=C.sizeof(rt_tune_info))


# numbering of rt_abi_sizeof / rt_abi_offsetof
ABI_STRUCTS = (rt_hittable, rt_texture, rt_camera, rt_scene_info, rt_stats, rt_render_options, rt_scene_options, rt_tune_info)
