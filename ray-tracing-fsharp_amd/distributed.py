"""Row-interleaved sharding of one frame over the GPUs of a node, one process per GPU.

Image row r belongs to rank r mod world (adaptive sampling makes sky rows ~11/spp the cost of object rows, so
contiguous tiles would be badly balanced).  Pixels are independent given the (seed, global pixel, sample) streams, so the
only exchange is ONE gather of the per-rank PixelStats accumulators to rank 0 -- `torch.distributed.gather`, which is
RCCL over xGMI with the "nccl" backend and gloo on CPU.  Integer sums make the result independent of the sharding.

torch is plumbing here (device buffers, the stream handle, the process group); rendering goes through the C ABI.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Optional, Tuple

import numpy as np

from . import _abi as A
from ._lib import check, lib
from .raytracing import Camera, Scene


def shard_rows(rows: int, rank: int, world: int) -> Tuple[int, int, int]:
    """-> (row_first, row_stride, n_rows) of `rank`'s interleaved shard."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return rank, world, max(0, (rows - rank + world - 1) // world)


def render_shard_device(scene: Scene, camera: Camera, maxWidthCoord: int, maxHeightCoord: int, seed: int, device: int,
                        row_first: int, row_stride: int, n_rows: int, accum, stream: int = 0, counters: bool = False,
                        want_stats: bool = False):
    """rt_render_device into `accum` (a CUDA int32 torch tensor [>= n_rows, cols, 4]), enqueued on `stream`."""
    cam = camera.to_abi()
    st = A.rt_stats() if want_stats else None
    check(lib.rt_render_device(scene.handle, C.byref(cam), maxWidthCoord, maxHeightCoord, seed, device, row_first, row_stride, n_rows,
                               A.RT_RENDER_COUNTERS if counters else 0, C.c_void_p(accum.data_ptr()), None, C.c_void_p(stream),
                               C.byref(st) if want_stats else None))
    return st.as_dict() if want_stats else None


def gather_frame(local_accum, rows: int, cols: int, rank: int, world: int, group=None):
    """Gather the padded per-rank accumulators ([ceil(rows/world), cols, 4] int32 each) to rank 0 and de-interleave.
    Returns the [rows, cols, 4] frame on rank 0, None elsewhere."""
    import torch
    import torch.distributed as dist

    n_pad = (rows + world - 1) // world
    assert tuple(local_accum.shape) == (n_pad, cols, 4)
    if world == 1:
        return local_accum[:rows]
    src = local_accum
    if dist.get_backend(group) == "gloo" and local_accum.is_cuda:
        src = local_accum.cpu()  # gloo has no device gather: stage through host memory (CPU tests, one-GPU rehearsals)
    bufs = [torch.empty_like(src) for _ in range(world)] if rank == 0 else None
    dist.gather(src, bufs, dst=0, group=group)
    if rank != 0:
        return None
    frame = torch.empty((rows, cols, 4), dtype=local_accum.dtype, device=local_accum.device)
    for r in range(world):
        n = shard_rows(rows, r, world)[2]
        frame[r::world] = bufs[r][:n]
    return frame


def render_frame(scene: Scene, camera: Camera, maxWidthCoord: int, maxHeightCoord: int, seed: int = 0, *, rank: int = 0,
                 world: int = 1, device: Optional[int] = None, group=None,
                 render_fn: Optional[Callable[[int, int, int], np.ndarray]] = None):
    """One frame across `world` ranks.  Default path: HIP render into a CUDA tensor, RCCL gather.
    `render_fn(row_first, row_stride, n_rows) -> int32 [n_rows, cols, 4]` replaces the renderer in the CPU/gloo tests."""
    import torch

    rows, cols = 2 * maxHeightCoord + 1, 2 * maxWidthCoord + 1
    first, stride, n = shard_rows(rows, rank, world)
    n_pad = (rows + world - 1) // world
    if render_fn is None:
        dev = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        local = torch.zeros((n_pad, cols, 4), dtype=torch.int32, device=dev)
        render_shard_device(scene, camera, maxWidthCoord, maxHeightCoord, seed, dev.index, first, stride, n, local,
                            stream=torch.cuda.current_stream(dev).cuda_stream)
    else:
        local = torch.zeros((n_pad, cols, 4), dtype=torch.int32)
        local[:n] = torch.from_numpy(np.ascontiguousarray(render_fn(first, stride, n), dtype=np.int32))
    return gather_frame(local, rows, cols, rank, world, group)


def mean_pixels(frame_accum: np.ndarray) -> np.ndarray:
    """PixelStats.mean (Pixel.fs:103-108) over a gathered frame: integer division of the sums by Count."""
    a = np.asarray(frame_accum)
    return (a[..., 1:] // a[..., :1]).astype(np.uint8)
