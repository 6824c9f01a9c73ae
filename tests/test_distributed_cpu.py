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


def test_a_tile_s_place_in_the_rank_s_pixel_list_in_closed_form():
    """owned_pixels_below (rt_core.h; the render kernels' k_owned_list places every tile with it and rt_render sizes the
    rank's share with it) against a count over the tiles, for ragged frames, tiles larger than the frame and one-pixel tiles."""
    import ctypes as C
    from oracle_lib import twin
    fn = twin().twin_owned_pixels_below
    fn.restype, fn.argtypes = C.c_uint64, [C.c_uint32] * 6 + [C.c_uint64]
    for (w, h, tw, th, world) in [(64, 36, 16, 16, 2), (100, 75, 16, 16, 3), (800, 800, 16, 16, 8), (17, 33, 8, 4, 4), (33, 17, 1, 1, 5), (40, 30, 64, 64, 2),
                                  (50, 20, 7, 20, 3), (19, 23, 19, 5, 2)]:
        ntx, nty = -(-w // min(tw, w)), -(-h // min(th, h))
        for rank in range(world):
            sizes = np.zeros(ntx * nty + 1, dtype=np.int64)   # pixels of every tile of the rank, 0 for the others'
            for t in range(ntx * nty):
                tx, ty = t % ntx, t // ntx
                if t % world == rank:
                    sizes[t] = min(tw, w - tx * tw) * min(th, h - ty * th)
            below = np.concatenate([[0], np.cumsum(sizes)])
            for t_end in list(range(0, ntx * nty + 1, max(1, ntx * nty // 37))) + [ntx * nty, ntx * nty + 5]:
                assert fn(w, h, tw, th, rank, world, t_end) == below[min(t_end, ntx * nty)], (w, h, tw, th, world, rank, t_end)
            assert fn(w, h, tw, th, rank, world, ntx * nty) == owned_mask(w, h, rank, world, min(tw, w), min(th, h)).sum()


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
