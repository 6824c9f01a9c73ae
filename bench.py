"""Benchmark of the path-tracing hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

One step = one full frame of the workload BASELINE.json quotes its metric on: the reference's
final_scene example at 800x800, 10 000 samples per pixel, 50 reflections, tile-split over the N
ranks and gathered on rank 0 (one RCCL sum-reduce of the f64 frame).  The scene is resident on
the device before the timed region; warm-up steps render the same frame at 128 spp (they size the
workspace).

Rank 0 prints one JSON line.  `roofline` prices the dominant kernel (BVH traversal) against the
fp32 vector peak: algorithmic flops per path segment are counted by the CPU oracle on the
reference's traversal order (per-operation constants below) and divided by the kernel's average
launch duration, measured here with HIP events in a profiled render after the timed one.  The
timed render overlaps the traversal of one half of a pass with the shading of the other half on
two streams; the profiled render (like --one-stream) runs the kernels one after the other, which
is the duration a kernel trace of `bench.py --one-stream` shows.  `cpu_baseline` is the CPU oracle (an f64
restatement of the reference, not the Rust binary) timed on this host on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np
import torch
import torch.distributed as dist

import raytracing_amd as rt
from raytracing_amd import _abi
from raytracing_amd.distributed import render_distributed

WIDTH = HEIGHT = 800
SPP = 10000
DEPTH = 50
PEAK_FP32_TFLOPS = 157.3  # MI355X fp32 vector peak

# flops per operation, read off the reference's source lines (one per add/sub/mul/div/min/max/compare/sqrt/transcendental)
TRAVERSE_COST = {"slab": 24, "sphere_miss": 24, "sphere_range": 30, "sphere_hit": 73, "rect_treject": 4, "rect_breject": 15, "rect_hit": 39,
                 "translate": 6, "rotate_miss": 20, "rotate_hit": 47, "medium": 45, "moving": 8}
SHADE_COST = {"lambertian": 63, "metal": 54, "dielectric": 62, "isotropic": 19, "tex_checker": 9, "tex_noise": 170, "tex_image": 17}


def algorithmic_flops(cnt):
    trav = sum(cnt[k] * c for k, c in TRAVERSE_COST.items())
    shade = sum(cnt[k] * c for k, c in SHADE_COST.items()) + 3 * cnt["segments"]
    raygen = 45 * cnt["samples"]
    return trav, shade, raygen


def hbm_traffic_per_launch(kernel, segments_per_launch):
    """HBM bytes of one launch of `kernel`: the bytes per path segment measured with the rocprofv3 PMC counters (kept with
    their derivation in profiles/) times the segments of this run's average launch.  None without that file."""
    files = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_hbm_traffic.json")) if os.path.isdir(os.path.join(ROOT, "profiles")) else []
    if not files:
        return None
    with open(os.path.join(ROOT, "profiles", files[-1])) as fh:
        per_segment = json.load(fh)["kernels"][kernel]["bytes_per_segment"]
    return round(per_segment * segments_per_launch)


def cpu_baseline_and_work(earth, timed):
    """Counts the reference algorithm's work per sample with the oracle (96x96, 8 spp) and, if `timed`, times the oracle on
    final_scene at 320x320, 96 spp (9.8 M samples of the full view, 10-30 s on 16 cores)."""
    from oracle_lib import OracleScene
    threads = min(16, os.cpu_count() or 1)
    osc, oco = OracleScene.example("final_scene", 1, earth)
    _, _, _, cnt = osc.render(oco, 96, 96, 8, DEPTH, seed=1, threads=threads, counters=True)
    if not timed:
        return None, cnt
    w = h = 320
    spp = 96
    _, _, secs, _ = osc.render(oco, w, h, spp, DEPTH, seed=1, threads=threads, shade_every=True)
    base = {"value": round(w * h * spp / secs / 1e6, 4), "unit": "Msamples/s", "cores": threads, "kind": "port",
            "sample": f"final_scene {w}x{h}, {spp} spp, depth {DEPTH}: {w * h * spp} samples in {secs:.1f} s; f64 C++ restatement of the reference "
                      f"(reference traversal order, every accepted candidate shaded), {threads} std::threads"}
    return base, cnt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp", type=int, default=SPP, help="samples per pixel of a timed step (default: the BASELINE configuration)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true", help="N > 1 ranks that all use GPU 0 and gather over gloo (functional rehearsal only)")
    ap.add_argument("--one-stream", action="store_true", help="render every pass as one half on one stream (RT_RENDER_ONE_STREAM): kernels do not "
                    "overlap, so a kernel trace shows the per-kernel durations that the roofline is quoted on")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.rehearse_on_one_gpu:
        local_rank = 0  # every rank on GPU 0, frames gathered over gloo: exercises the N > 1 path on a one-GPU box (not a measurement)
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    earth = rt.earthmap()
    scene, co = rt.Scene.example("final_scene", 1, earth)
    cam = rt.Camera(co)
    scene.device(local_rank)
    io = rt.ImageOptions(HEIGHT, WIDTH, args.spp, DEPTH)
    io_warm = rt.ImageOptions(HEIGHT, WIDTH, min(128, args.spp), DEPTH)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    flags = _abi.RT_RENDER_ONE_STREAM if args.one_stream else 0
    for _ in range(args.warmup):
        render_distributed(scene, cam, io_warm, seed=1, flags=flags)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rgb, mean = render_distributed(scene, cam, io, seed=1, flags=flags)
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.rehearse_on_one_gpu else "cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    result = None
    if rank == 0:
        samples = WIDTH * HEIGHT * args.spp * args.steps
        value = samples / elapsed / 1e6
        # dominant-kernel timing: a profiled render (HIP events around every launch, on the launch stream) of this rank's share
        # (104 spp = one whole pass of 64 M paths at 800x800 on one GPU: every launch of the profile is a full-size one, and a kernel
        # trace of `bench.py --spp 104 --one-stream` averages over launches of exactly this kind)
        prof_io = rt.ImageOptions(HEIGHT, WIDTH, min(104, args.spp), DEPTH)
        dmean = torch.empty((HEIGHT, WIDTH, 3), dtype=torch.float64, device="cuda")
        scene.render_device(cam, prof_io, rt.RenderOptions(seed=1, rank=0, world_size=world, tile_w=16, tile_h=16, flags=_abi.RT_RENDER_PROFILE), dmean.data_ptr())
        torch.cuda.synchronize()
        prof = scene.profile()
        roofline, cpu = None, None
        if not args.no_cpu_baseline:
            cpu, cnt = cpu_baseline_and_work(earth, timed=world == 1)
            trav, shade, raygen = algorithmic_flops(cnt)
            trav_per_seg = trav / cnt["segments"]
            seg_per_launch = prof["traverse_segments"] / max(prof["traverse_launches"], 1)
            ms_per_launch = prof["traverse_ms"] / max(prof["traverse_launches"], 1)
            achieved = trav_per_seg * seg_per_launch / (ms_per_launch * 1e-3) / 1e12
            roofline = {"bound": "valu", "kernel": "k_traverse", "achieved": round(achieved, 3), "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(achieved / PEAK_FP32_TFLOPS, 4), "traffic": hbm_traffic_per_launch("k_traverse", seg_per_launch),
                        "flops_per_segment": round(trav_per_seg, 1), "segments_per_launch": round(seg_per_launch), "ms_per_launch": round(ms_per_launch, 4),
                        "flops_per_sample_all_kernels": round((trav + shade + raygen) / cnt["samples"], 1),
                        "segments_per_sample": round(cnt["segments"] / cnt["samples"], 2),
                        "kernel_ms": {k: round(prof[k + "_ms"], 2) for k in _abi.RT_K_NAMES}}
        result = {"metric": "Msamples/sec (whole node) on final_scene", "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world,
                  "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
                  "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                  "config": {"workload": f"final_scene {WIDTH}x{HEIGHT}, {args.spp} spp, depth {DEPTH}, interleaved 16x16 tiles over {world} GPU(s), gather to GPU0",
                             "scene_seed": 1, "render_seed": 1, "objects": 1410},
                  "roofline": roofline, "cpu_baseline": cpu}
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return result


if __name__ == "__main__":
    main()
