"""The N > 1 path on CPU: two gloo ranks split a frame into interleaved tiles, each fills only its own
pixels (here with the CPU build of the kernel code standing in for the GPU render), and the frames
are sum-reduced to rank 0 exactly as raytracing_amd.distributed does with RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracing_amd.distributed import gather_frames, owned_mask


def test_tiles_partition_the_frame():
    for (w, h, world) in [(64, 36, 2), (100, 75, 3), (800, 800, 8), (17, 33, 4)]:
        masks = [owned_mask(w, h, r, world) for r in range(world)]
        total = np.sum(masks, axis=0)
        assert (total == 1).all()                       # every pixel owned by exactly one rank
        share = [m.mean() for m in masks]
        if w * h >= 64 * 36:
            assert max(share) - min(share) < 0.5 / world + 0.1   # interleaving keeps the shares close


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import raytracing_amd as rt
    from oracle_lib import twin_render
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc, co = rt.Scene.example("cornell_smoke", 1)
    cam = rt.Camera(co)
    w, h = 40, 24
    full = twin_render(sc, cam, w, h, 4, 50, seed=1, threads=2)
    mine = torch.from_numpy(np.where(owned_mask(w, h, rank, world)[:, :, None], full, 0.0).copy())
    gather_frames(mine, 0)
    if rank == 0:
        q.put(bool(np.array_equal(mine.numpy(), full)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_is_bit_identical():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok
