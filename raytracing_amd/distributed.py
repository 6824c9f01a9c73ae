"""Multi-GPU rendering: one process per GPU, frames split into interleaved tiles.

Every pixel is rendered wholly by the rank that owns its tile (tile_id % world_size == rank),
with Philox streams keyed on the pixel, so a pixel's value does not depend on how many ranks
take part.  The only exchange is one sum-reduce of the full-frame f64 mean-radiance buffers to
rank 0 (RCCL over xGMI when the process group's backend is "nccl"): every rank holds zeros for
the pixels it does not own, and adding zeros is exact, so the gathered frame is bit-identical to a
single-rank render.

Stream order.  The library launches on the stream it is given; a NULL handle means its own
non-blocking stream, which has no order with torch's streams -- and the handle of torch's default
stream IS 0.  The render and the quantisation therefore run on a torch side stream of this module
(one per device, a real handle), joined to torch's current stream by events on both sides, so
that the collective (ordered after the current stream) reads a finished frame and the caller only
ever has to synchronise the current stream.
"""
import os

import numpy as np
import torch
import torch.distributed as dist

from .api import Camera, ImageOptions, RenderOptions, Scene

TILE_W = TILE_H = 16

_side_streams = {}


def render_stream(device: torch.device) -> torch.cuda.Stream:
    """The side stream renders run on (one per device, created on first use)."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx not in _side_streams:
        _side_streams[idx] = torch.cuda.Stream(device=idx)
    return _side_streams[idx]


def owned_mask(width, height, rank, world_size, tile_w=TILE_W, tile_h=TILE_H):
    """(height, width) bool array in output orientation (row 0 = top): pixels of this rank's tiles.
    Tiles are counted over (column i, row j from the bottom) as the library does: tile_id = ty * ntx + tx."""
    ntx = (width + tile_w - 1) // tile_w
    i = np.arange(width)[None, :]
    j = (height - 1 - np.arange(height))[:, None]
    tile_id = (j // tile_h) * ntx + (i // tile_w)
    return (tile_id % world_size) == rank


def gather_frames(local_mean: torch.Tensor, dst: int = 0, group=None) -> torch.Tensor:
    """Sum-reduce of the per-rank frames (zeros outside the owned tiles) to `dst`; in place.  Runs whenever a process group
    exists, a group of one included (the collective is then the identity, and a one-GPU box can exercise the RCCL call)."""
    if dist.is_available() and dist.is_initialized():
        if local_mean.is_cuda and dist.get_backend(group) == "gloo" and not os.environ.get("RT_GLOO_DEVICE_TENSORS"):
            # rehearsals only (several ranks on one GPU, no RCCL): the frame goes through the host here (a copy on the current
            # stream, which waits for the frame), and gloo sums host tensors.  Handed the device tensor, gloo stages it through
            # pinned memory on a HIGH-PRIORITY stream of torch's pool that it makes wait for the current stream; on this ROCm
            # build a raised-priority stream waiting for normal-priority streams that still have work (the frame being rendered)
            # is never released: its hipStreamSynchronize in gloo's worker thread does not return (tools/gloo_hang_replica.py
            # shows it in one process without gloo; the same chain through a stream of priority 0 completes).  RCCL's process
            # group uses streams of normal priority.
            host = local_mean.cpu()
            dist.reduce(host, dst=dst, op=dist.ReduceOp.SUM, group=group)
            if dist.get_rank(group) == dst:
                local_mean.copy_(host)
        else:
            dist.reduce(local_mean, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return local_mean


def render_local(scene: Scene, camera: Camera, image_options: ImageOptions, seed: int = 1, rank: int = 0, world_size: int = 1, flags: int = 0) -> torch.Tensor:
    """This rank's tiles as a full frame of f64 means (zeros elsewhere) on the current device.  The work runs on the
    module's side stream; on return torch's current stream is ordered after it, so the tensor can be used there
    (and by collectives) without a device-wide synchronisation."""
    dev = torch.device("cuda", torch.cuda.current_device())
    h, w = image_options.height, image_options.width
    cur, side = torch.cuda.current_stream(dev), render_stream(dev)
    mean = torch.empty((h, w, 3), dtype=torch.float64, device=dev)
    side.wait_stream(cur)          # (the allocator may hand out memory that work queued on `cur` still uses)
    mean.record_stream(side)
    ro = RenderOptions(seed=seed, rank=rank, world_size=world_size, tile_w=TILE_W, tile_h=TILE_H, flags=flags, stream=side.cuda_stream)
    scene.render_device(camera, image_options, ro, mean.data_ptr())
    cur.wait_stream(side)
    return mean


def quantise(scene: Scene, mean: torch.Tensor) -> torch.Tensor:
    """lib.rs:68-72 on a device frame of means: uint8 tensor of the same shape, ordered on torch's current stream."""
    dev = mean.device
    cur, side = torch.cuda.current_stream(dev), render_stream(dev)
    rgb = torch.empty(mean.shape, dtype=torch.uint8, device=dev)
    side.wait_stream(cur)          # the frame may still be being reduced on `cur`
    rgb.record_stream(side)
    mean.record_stream(side)
    scene.quantise_device(mean.data_ptr(), mean.numel(), rgb.data_ptr(), side.cuda_stream)
    cur.wait_stream(side)
    return rgb


def render_distributed(scene: Scene, camera: Camera, image_options: ImageOptions, seed: int = 1, flags: int = 0, group=None):
    """Renders this rank's tiles on the current CUDA/HIP device, gathers on rank 0.
    Returns (rgb uint8 (h, w, 3) device tensor, mean f64 device tensor) on rank 0 and (None, local mean) elsewhere.
    Everything is ordered on torch's current stream when this returns; the caller synchronises that stream."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    mean = render_local(scene, camera, image_options, seed, rank, world, flags)
    gather_frames(mean, 0, group)
    if rank != 0:
        return None, mean
    return quantise(scene, mean), mean
