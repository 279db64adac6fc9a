"""The N>1 path on CPU: two processes, gloo backend, row-interleaved shards, one gather to rank 0
(ray_tracing_fsharp_amd.distributed).  The per-rank renderer is the oracle here (render_fn), because this container has
no GPU; on the GPU box the same code path runs with the HIP renderer and RCCL."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    import oracle as orc
    import ray_tracing_fsharp_amd as rt
    import scenes
    from ray_tracing_fsharp_amd import distributed as rtd

    dist.init_process_group("gloo", rank=rank, world_size=world)
    objs, cam, w, h = scenes.all_materials(pixels=9)  # 19 rows: an odd count, so the shards are ragged
    o = orc.OracleScene(objs)

    def render_fn(first, stride, n):
        return o.render_rows(w, h, cam.to_abi(), seed=21, row_first=first, row_stride=stride, n_rows=n, threads=2)[0]

    frame = rtd.render_frame(rt.Scene.make(objs), cam, w, h, seed=21, rank=rank, world=world, render_fn=render_fn)
    if rank == 0:
        np.save(out_path, frame.numpy())
    else:
        assert frame is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_two_rank_gloo_frame_equals_single_process(rt, orc, tmp_path, world):
    import torch.multiprocessing as mp

    import scenes

    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    frame = np.load(out)
    objs, cam, w, h = scenes.all_materials(pixels=9)
    full = orc.OracleScene(objs).render_rows(w, h, cam.to_abi(), seed=21, threads=4)[0]
    assert frame.shape == full.shape and np.array_equal(frame, full)
    from ray_tracing_fsharp_amd import distributed as rtd
    assert np.array_equal(rtd.mean_pixels(frame), (full[..., 1:] // full[..., :1]).astype(np.uint8))


def test_shard_rows_partition_the_frame(rt):
    from ray_tracing_fsharp_amd import distributed as rtd

    for rows in (1, 7, 101, 1601):
        for world in (1, 2, 3, 8, 16):
            seen = []
            for r in range(world):
                first, stride, n = rtd.shard_rows(rows, r, world)
                seen += list(range(first, rows, stride))[:n]
                assert n == len(range(first, rows, stride))
            assert sorted(seen) == list(range(rows))
    with pytest.raises(ValueError):
        rtd.shard_rows(10, 2, 2)
