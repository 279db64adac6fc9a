"""ray-tracing-fsharp_amd: the MI355X-native per-pixel sampling path of Smaug123/ray-tracing-fsharp.

`csrc/` holds the hand-written HIP path and the C ABI (include/rtfs_amd.h); the modules here are the host-side
mirror of the reference's construction/render API over that ABI.  Import fails loudly if librtfs_amd.so is missing.
"""
from . import _abi  # noqa: F401
from ._lib import LIB_PATH, RtError, lib  # noqa: F401
from .raytracing import (Camera, Colour, FloatProducer, Hittable, Image, ImageOutput, InfinitePlane, InfinitePlaneStyle,  # noqa: F401
                         ParameterisedTexture, Pixel, PixelOutput, Point, RenderResult, Scene, Sphere, SphereStyle, Texture,
                         UnitVector, Vector)
from . import hooks, sample_images  # noqa: F401


def device_count() -> int:
    return int(lib.rt_device_count())


def set_launch_config(block_threads: int = 0, chunk_pixels: int = 0, blocks_per_cu: int = 0) -> None:
    from ._lib import check
    check(lib.rt_set_launch_config(block_threads, chunk_pixels, blocks_per_cu))


def set_schedule(yield_lanes: int = 0, refill_lanes: int = 0) -> None:
    from ._lib import check
    check(lib.rt_set_schedule(yield_lanes, refill_lanes))


def set_passes(passes: int = 0) -> None:
    """0 auto, 1 fused kernel, 2 two-pass (phase 1 + decision, then cost-ordered phase 2).  Never changes a result."""
    from ._lib import check
    check(lib.rt_set_passes(passes))


def set_park(park_lanes: int = 0) -> None:
    """Capacity of a wave's pool of parked rare-style paths: 0 default, -1 never park.  Never changes a result."""
    from ._lib import check
    check(lib.rt_set_park(park_lanes))


def set_walk_tree(kind="sah") -> None:
    """Which tree over the Leaf boxes scenes created AFTERWARDS hand to the device: "sah" (default; fewer box tests per ray) or
    "reference" (BoundingBoxTree.make's own; its box-test count equals the reference's).  Pixels are identical under both."""
    from . import _abi as A
    from ._lib import check
    global _walk_tree
    k = {"sah": A.RT_WALK_TREE_SAH, "reference": A.RT_WALK_TREE_REFERENCE}.get(kind, kind)
    check(lib.rt_set_walk_tree(int(k)))
    _walk_tree = "reference" if int(k) == A.RT_WALK_TREE_REFERENCE else "sah"


_walk_tree = "sah"


def get_walk_tree() -> str:
    return _walk_tree
