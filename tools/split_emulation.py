"""How evenly the interleaved-tile split loads the ranks, measured on one GPU: for N in {2, 4, 8} every rank's share of the
final_scene frame (rt_render_device with rank r of N) is rendered back to back and timed; the predicted N-GPU frame time is the
slowest rank's (the gather of a 15 MB frame over xGMI is microseconds against seconds of rendering).

    python tools/split_emulation.py [spp] [tile] > profiles/<tag>_split_emulation.md

Each share is first rendered at the sample count of one full pass of the share plus a little (as bench.py's warm-up does: sizes
the workspace, lets the library learn the schedule of passes of that size), then timed at the full sample count."""
import sys
import time
sys.path.insert(0, '.')
import torch
import raytracing_amd as rt

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
tile = int(sys.argv[2]) if len(sys.argv) > 2 else 16
W = H = 800
sc, co = rt.Scene.example("final_scene", 1, rt.earthmap())
cam = rt.Camera(co)
buf = torch.empty((H, W, 3), dtype=torch.float64, device="cuda")


def timed(rank, world):
    from raytracing_amd import _abi
    from raytracing_amd.distributed import owned_mask
    n_owned = int(owned_mask(W, H, rank, world, tile, tile).sum())
    warm = min(spp, int(_abi.hip_lib().rt_auto_pass_paths()) // max(n_owned, 1) + 24)   # one full pass of this share and a little, as bench.py warms up
    for s in (warm, spp):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sc.render_device(cam, rt.ImageOptions(H, W, s, 50), rt.RenderOptions(seed=1, rank=rank, world_size=world, tile_w=tile, tile_h=tile), buf.data_ptr())
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return dt


t1 = timed(0, 1)
print(f"# Tile-split emulation on one MI355X: final_scene {W}x{H}, {spp} spp, {tile}x{tile} tiles, owner = tile_id % N\n")
print(f"One rank renders the whole frame in {t1:.3f} s ({W * H * spp / t1 / 1e6:.1f} Msamples/s).  Each row: the N shares timed one after the other on the same GPU.\n")
print("| N | slowest share (s) | mean share (s) | max / mean | sum of shares / T1 | predicted speed-up T1 / max | predicted Msamples/s |")
print("|---|---|---|---|---|---|---|")
for n in (2, 4, 8):
    ts = [timed(r, n) for r in range(n)]
    mx, mean = max(ts), sum(ts) / n
    print(f"| {n} | {mx:.3f} | {mean:.3f} | {mx / mean:.3f} | {sum(ts) / t1:.3f} | {t1 / mx:.2f} | {W * H * spp / mx / 1e6:.0f} |", flush=True)
    print(f"<!-- N={n} shares: {' '.join(f'{t:.3f}' for t in ts)} -->")
print("\nmax / mean is the load imbalance of the split; sum of shares / T1 above 1 is what a smaller share loses to fixed costs "
      "(shorter passes, the same number of launches).  No multi-GPU node was available to this build: these are predictions from one GPU, not a scaling curve.")
