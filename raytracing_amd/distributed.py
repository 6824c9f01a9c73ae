"""Multi-GPU rendering: one process per GPU, frames split into interleaved tiles.

Every pixel is rendered wholly by the rank that owns its tile (tile_id % world_size == rank),
with Philox streams keyed on the pixel, so a pixel's value does not depend on how many ranks
take part.  The only exchange is one sum-reduce of the full-frame f64 mean-radiance buffers to
rank 0 (RCCL over xGMI when the process group's backend is "nccl"): every rank holds zeros for
the pixels it does not own, and adding zeros is exact, so the gathered frame is bit-identical to a
single-rank render.
"""
import numpy as np
import torch
import torch.distributed as dist

from .api import Camera, ImageOptions, RenderOptions, Scene

TILE_W = TILE_H = 16


def owned_mask(width, height, rank, world_size, tile_w=TILE_W, tile_h=TILE_H):
    """(height, width) bool array in output orientation (row 0 = top): pixels of this rank's tiles.
    Tiles are counted over (column i, row j from the bottom) as the library does: tile_id = ty * ntx + tx."""
    ntx = (width + tile_w - 1) // tile_w
    i = np.arange(width)[None, :]
    j = (height - 1 - np.arange(height))[:, None]
    tile_id = (j // tile_h) * ntx + (i // tile_w)
    return (tile_id % world_size) == rank


def gather_frames(local_mean: torch.Tensor, dst: int = 0, group=None) -> torch.Tensor:
    """Sum-reduce of the per-rank frames (zeros outside the owned tiles) to `dst`; in place."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if local_mean.is_cuda and dist.get_backend(group) == "gloo":
            dist.all_reduce(local_mean, op=dist.ReduceOp.SUM, group=group)  # gloo has no reduce() for device tensors (rehearsals only)
        else:
            dist.reduce(local_mean, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return local_mean


def render_distributed(scene: Scene, camera: Camera, image_options: ImageOptions, seed: int = 1, flags: int = 0, group=None):
    """Renders this rank's tiles on the current CUDA/HIP device, gathers on rank 0.
    Returns (rgb uint8 (h, w, 3) device tensor, mean f64 device tensor) on rank 0 and (None, local mean) elsewhere.
    Everything is enqueued on torch's current stream; the caller synchronises."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    dev = torch.device("cuda", torch.cuda.current_device())
    h, w = image_options.height, image_options.width
    mean = torch.empty((h, w, 3), dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    ro = RenderOptions(seed=seed, rank=rank, world_size=world, tile_w=TILE_W, tile_h=TILE_H, flags=flags, stream=stream)
    scene.render_device(camera, image_options, ro, mean.data_ptr())
    gather_frames(mean, 0, group)
    if rank != 0:
        return None, mean
    rgb = torch.empty((h, w, 3), dtype=torch.uint8, device=dev)
    scene.quantise_device(mean.data_ptr(), 3 * h * w, rgb.data_ptr(), stream)
    return rgb, mean
