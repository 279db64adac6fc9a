"""GPU tests of the boundary itself (include/rtfs_amd.h): per-call options instead of process-wide setters, any number of
unsynchronised launches in flight, the caller's current device left alone, the two-pass unit sizes at the edge of the LDS, and
rt_render_frame (one process, several devices, one gather)."""
import ctypes as C
import dataclasses
import os

import numpy as np
import pytest

import scenes

pytestmark = pytest.mark.gpu


def test_more_than_64_unsynchronised_launches_each_keep_their_own_scratch(rt):
    """rt_render_device with stats == NULL returns right after the launch.  Earlier versions took a launch's counters, work queue
    and camera from a 64-slot ring, so the 65th outstanding launch reused the slot of one that might still be running; now every
    launch owns stream-ordered scratch.  100 launches with four different cameras on one stream, none synchronised until the end."""
    import torch

    from ray_tracing_fsharp_amd import distributed as rtd

    objs, cam, w, h = scenes.small_final(spp=24, pixels=8)
    s = rt.Scene.make(objs)
    cams = [dataclasses.replace(cam, SamplesPerPixel=spp, BounceDepth=depth) for spp, depth in ((24, 50), (12, 3), (30, 9), (1, 50))]
    want = [s.render_rows(w, h, c, seed=7 + i).accum for i, c in enumerate(cams)]
    rows, cols = 2 * h + 1, 2 * w + 1
    bufs = [torch.zeros((rows, cols, 4), dtype=torch.int32, device="cuda:0") for _ in range(100)]
    stream = torch.cuda.current_stream().cuda_stream
    for i, b in enumerate(bufs):
        rtd.render_shard_device(s, cams[i % 4], w, h, 7 + i % 4, 0, 0, 1, rows, b, stream=stream)
    torch.cuda.synchronize()
    for i, b in enumerate(bufs):
        assert np.array_equal(b.cpu().numpy(), want[i % 4]), i


def test_launches_on_several_streams_at_once(rt):
    import torch

    from ray_tracing_fsharp_amd import distributed as rtd

    objs, cam, w, h = scenes.all_materials(pixels=10)
    s = rt.Scene.make(objs)
    want = s.render_rows(w, h, cam, seed=3).accum
    rows, cols = 2 * h + 1, 2 * w + 1
    streams = [torch.cuda.Stream() for _ in range(6)]
    bufs = [torch.zeros((rows, cols, 4), dtype=torch.int32, device="cuda:0") for _ in range(36)]
    torch.cuda.synchronize()
    for i, b in enumerate(bufs):
        rtd.render_shard_device(s, cam, w, h, 3, 0, 0, 1, rows, b, stream=streams[i % 6].cuda_stream)
    torch.cuda.synchronize()
    assert all(np.array_equal(b.cpu().numpy(), want) for b in bufs)


def test_per_call_options_override_nothing_global(rt):
    """rt_render_device_ex: block size, unit size, passes, park pool and thresholds travel with the call; the process-wide
    defaults stay untouched and every setting gives the same integers."""
    import torch

    from ray_tracing_fsharp_amd import _abi as A
    from ray_tracing_fsharp_amd._lib import check, lib

    objs, cam, w, h = scenes.small_final(spp=30, pixels=10)
    s = rt.Scene.make(objs)
    base = s.render_rows(w, h, cam, seed=4, counters=True)
    rows, cols = 2 * h + 1, 2 * w + 1
    camabi = cam.to_abi()
    for kw in (dict(block_threads=256, chunk_pixels=8), dict(block_threads=512, passes=2), dict(passes=1, park_lanes=-1),
               dict(park_lanes=8, yield_lanes=20, refill_lanes=30), dict(block_threads=768, chunk_pixels=64, passes=2, park_lanes=256)):
        opt = A.rt_render_options(**kw)
        out = torch.zeros((rows, cols, 4), dtype=torch.int32, device="cuda:0")
        st = A.rt_stats()
        check(lib.rt_render_device_ex(s.handle, C.byref(camabi), w, h, 4, 0, 0, 1, rows, A.RT_RENDER_COUNTERS, C.c_void_p(out.data_ptr()), None,
                                      None, C.byref(opt), C.byref(st)))
        assert np.array_equal(out.cpu().numpy(), base.accum), kw
        assert st.rays == base.stats["rays"] and st.samples == base.stats["samples"], kw
    bad = A.rt_render_options(block_threads=100)
    assert lib.rt_render_device_ex(s.handle, C.byref(camabi), w, h, 4, 0, 0, 1, rows, 0, C.c_void_p(out.data_ptr()), None, None, C.byref(bad), None) == A.RT_ERR_INVALID_ARGUMENT
    assert s.info()["lds_resident"] == 1


def test_scene_options_pick_the_walk_tree_per_scene(rt):
    objs, cam, w, h = scenes.small_final(spp=12, pixels=8)
    a, b = rt.Scene.make(objs, walk_tree="sah"), rt.Scene.make(objs, walk_tree="reference")
    assert (a.info()["walk_tree"], b.info()["walk_tree"]) == (0, 1) and rt.get_walk_tree() in ("sah", "reference")
    ra, rb = a.render_rows(w, h, cam, seed=2, counters=True), b.render_rows(w, h, cam, seed=2, counters=True)
    assert np.array_equal(ra.accum, rb.accum) and ra.stats["aabb_tests"] < rb.stats["aabb_tests"] and ra.stats["prim_tests"] == rb.stats["prim_tests"]


def test_two_pass_unit_sizes_at_the_edge_of_the_lds(rt):
    """Scenes whose LDS image leaves room for 16-pixel units but not for pass B's 32-pixel ones (the bench scene itself sits in
    that band: 485 spheres): the unit size is settled before anything is launched, and two passes equal the fused kernel."""
    for n in (490, 520, 550):  # 490, 520: 16-pixel units fit, 32 do not; 550: only the 256-thread block keeps the scene in LDS
        objs, cam, w, h = scenes.many_spheres(n=n, seed=6, spp=14, depth=8, pixels=480)  # 1.6 Mpx: pass B would ask for 32-pixel units
        s = rt.Scene.make(objs)
        if not (os.environ.get("RTFS_BLOCK") or os.environ.get("RTFS_CHUNK")):  # (the stress knobs of conftest.py change what fits)
            assert s.info()["lds_resident"] == 1, n
        try:
            rt.set_passes(1)
            fused = s.render_rows(w, h, cam, seed=9, counters=True)
            rt.set_passes(2)
            two = s.render_rows(w, h, cam, seed=9, counters=True)
        finally:
            rt.set_passes(0)
        assert np.array_equal(fused.accum, two.accum) and fused.stats["rays"] == two.stats["rays"], n


def test_render_frame_one_process_several_devices(rt):
    """rt_render_frame: rows interleaved over the device list, one gather, frame handed back de-interleaved.  One GPU here, so
    the list repeats device 0 (peer-copy and host gathers), and the RCCL leg runs as a self-send on a one-rank communicator
    (ncclCommInitAll, ncclGroupStart, ncclSend + ncclRecv, ncclGroupEnd on the real library)."""
    from ray_tracing_fsharp_amd import _abi as A

    objs, cam, w, h = scenes.all_materials(pixels=11)  # 23 rows: ragged over 2, 3, 4 devices
    s = rt.Scene.make(objs)
    want = s.render_rows(w, h, cam, seed=12, counters=True)
    for devices, gather in (((0,), A.RT_GATHER_AUTO), ((0,), A.RT_GATHER_HOST), ((0, 0), A.RT_GATHER_PEER), ((0, 0, 0), A.RT_GATHER_HOST),
                            ((0, 0, 0, 0), A.RT_GATHER_AUTO), ((0,), A.RT_GATHER_RCCL)):
        got = s.render_frame(w, h, cam, seed=12, devices=devices, gather=gather, counters=True)
        assert np.array_equal(got.accum, want.accum) and np.array_equal(got.rgb, want.rgb), (devices, gather)
        for k in ("rays", "prim_tests", "reflections", "samples", "pixels", "pixels_early"):
            assert sum(st[k] for st in got.stats) == want.stats[k], (devices, gather, k)
    with pytest.raises(rt.RtError):
        s.render_frame(w, h, cam, seed=12, devices=(0, 0), gather=A.RT_GATHER_RCCL)  # one communicator rank per GPU
    with pytest.raises(rt.RtError):
        s.render_frame(w, h, cam, seed=12, devices=(5,))


def test_the_callers_current_device_is_left_alone(rt):
    import torch

    objs, cam, w, h = scenes.all_materials(pixels=4)
    before = torch.cuda.current_device()
    rt.Scene.make(objs).render_rows(w, h, cam, seed=1, device=0)
    assert torch.cuda.current_device() == before
