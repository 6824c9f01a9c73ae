"""Benchmark of the path-tracing hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config NAME]     (N > 1: launched by torch.distributed.run)

One step = one full frame of a BASELINE configuration, tile-split over the N ranks, gathered on rank 0 (one RCCL
sum-reduce of the f64 frame), quantised there and copied to the host: the metric runs to image-on-host.  The default
configuration is the one BASELINE.json quotes its metric on: the reference's final_scene example at 800x800, 10 000
samples per pixel, 50 reflections.  --config picks one of the other GPU configurations (cornell 600x600 at 1 000 spp,
perlin 1920x1080 at 1 000 spp, cornell_smoke 800x800 at 4 000 spp).  The scene is resident on the device before the
timed region; warm-up steps render the same frame at 128 spp (they size the workspace and let the library learn the
frame's schedule).

Rank 0 prints one JSON line.  `roofline` prices the kernel that takes most of the frame.  For the BVH traversal that is
the fp32 vector peak: algorithmic flops per path segment are counted by the CPU oracle on the reference's traversal order
(per-operation constants below).  For shading it is the HBM peak: 56 bytes read per segment (ray, throughput, hit record)
and 48 written when the path goes on or 16 when it ends (scenes whose bounce is one kernel: no hit record, and no read in
a path's first segment, whose camera ray the kernel makes itself).  Either is divided by the kernel's average launch duration,
measured here with HIP events in a profiled render after the timed one (kernels one after the other on one stream, which is
the duration a kernel trace of `bench.py --one-stream` shows; the timed render overlaps the traversal of one half of a pass
with the shading of the other half on two streams where that pays).  `roofline.counters` are the rocprofv3 PMC figures of
the same kernel on the same scene (VALU busy, L2 hit rate, HBM bytes per segment), read from the newest profiles/*_pmc.json;
`traffic` is that file's bytes per segment times this run's segments per launch.  `cpu_baseline` is the CPU oracle (an f64
restatement of the reference, not the Rust binary) timed on this host on a bounded sample of the same frame.
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

DEPTH = 50
# name: (width, height, samples per pixel, objects) -- the four GPU configurations of BASELINE.json
CONFIGS = {"cornell": (600, 600, 1000, 8), "perlin": (1920, 1080, 1000, 4), "cornell_smoke": (800, 800, 4000, 8), "final_scene": (800, 800, 10000, 1410)}
PEAK_FP32_TFLOPS = 157.3  # MI355X fp32 vector peak
PEAK_HBM_GBPS = 8000.0    # MI355X HBM3E peak

# flops per operation, read off the reference's source lines (one per add/sub/mul/div/min/max/compare/sqrt/transcendental)
TRAVERSE_COST = {"slab": 24, "sphere_miss": 24, "sphere_range": 30, "sphere_hit": 73, "rect_treject": 4, "rect_breject": 15, "rect_hit": 39,
                 "translate": 6, "rotate_miss": 20, "rotate_hit": 47, "medium": 45, "moving": 8}
SHADE_COST = {"lambertian": 63, "metal": 54, "dielectric": 62, "isotropic": 19, "tex_checker": 9, "tex_noise": 170, "tex_image": 17}
# queue bytes of one shaded segment: ray + throughput + hit record in; the next ray out, or the sample's value
SHADE_BYTES_IN, SHADE_BYTES_OUT_ALIVE, SHADE_BYTES_OUT_DEAD = 56, 48, 16


def algorithmic_flops(cnt):
    trav = sum(cnt[k] * c for k, c in TRAVERSE_COST.items())
    shade = sum(cnt[k] * c for k, c in SHADE_COST.items()) + 3 * cnt["segments"]
    raygen = 45 * cnt["samples"]
    return trav, shade, raygen


def pmc_counters(config, kernel):
    """The rocprofv3 PMC figures of `kernel` on `config` from the newest profiles/*_pmc.json (made by tools/pmc_summarise.py), or None."""
    d = os.path.join(ROOT, "profiles")
    files = sorted(f for f in os.listdir(d) if f.endswith("_pmc.json")) if os.path.isdir(d) else []
    if not files:
        return None
    with open(os.path.join(d, files[-1])) as fh:
        doc = json.load(fh)
    rec = doc.get("configs", {}).get(config, {}).get("kernels", {}).get(kernel)
    if rec is None:
        return None
    out = {k: rec[k] for k in ("valu_busy", "l2_hit_rate", "hbm_bytes_per_segment", "hbm_gbps", "valu_lane_slots_per_segment") if k in rec}
    out["source"] = "profiles/" + files[-1]
    return out


def cpu_baseline_and_work(name, earth, timed):
    """Counts the reference algorithm's work per sample with the oracle (a 96-wide frame, 8 spp) and, if `timed`, times the
    oracle on a bounded sample of the configuration's frame: the full view at 2/5 of the size, the sample count chosen from a
    pilot render so that the timed render takes about 15 s on this host's cores."""
    from oracle_lib import OracleScene
    w, h, _, _ = CONFIGS[name]
    threads = min(16, os.cpu_count() or 1)
    osc, oco = OracleScene.example(name, 1, earth)
    cw = 96 if w == h else 128
    ch = cw * h // w
    _, _, _, cnt = osc.render(oco, cw, ch, 8, DEPTH, seed=1, threads=threads, counters=True)
    if not timed:
        return None, cnt
    sw, sh = w * 2 // 5, h * 2 // 5
    _, _, pilot, _ = osc.render(oco, sw, sh, 4, DEPTH, seed=1, threads=threads, shade_every=True)
    spp = int(max(8, min(1024, round(15.0 / max(pilot, 1e-3) * 4))))
    _, _, secs, _ = osc.render(oco, sw, sh, spp, DEPTH, seed=1, threads=threads, shade_every=True)
    base = {"value": round(sw * sh * spp / secs / 1e6, 4), "unit": "Msamples/s", "cores": threads, "kind": "port",
            "sample": f"{name} {sw}x{sh}, {spp} spp, depth {DEPTH}: {sw * sh * spp} samples in {secs:.1f} s; f64 C++ restatement of the reference "
                      f"(reference traversal order, every accepted candidate shaded), {threads} std::threads"}
    return base, cnt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="final_scene", help="BASELINE configuration to run (default: the one the metric is quoted on)")
    ap.add_argument("--spp", type=int, default=0, help="samples per pixel of a timed step (default: the configuration's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true", help="N > 1 ranks that all use GPU 0 and gather over gloo (functional rehearsal only)")
    ap.add_argument("--one-stream", action="store_true", help="render every pass as one half on one stream (RT_RENDER_ONE_STREAM): kernels do not "
                    "overlap, so a kernel trace shows the per-kernel durations that the roofline is quoted on")
    args = ap.parse_args()
    WIDTH, HEIGHT, SPP, n_objects = CONFIGS[args.config]
    spp = args.spp or SPP

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

    earth = rt.earthmap() if args.config == "final_scene" else None
    scene, co = rt.Scene.example(args.config, 1, earth)
    cam = rt.Camera(co)
    scene.device(local_rank)
    io = rt.ImageOptions(HEIGHT, WIDTH, spp, DEPTH)
    io_warm = rt.ImageOptions(HEIGHT, WIDTH, min(128, spp), DEPTH)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    flags = _abi.RT_RENDER_ONE_STREAM if args.one_stream else 0
    host_rgb = torch.empty((HEIGHT, WIDTH, 3), dtype=torch.uint8).pin_memory() if rank == 0 else None
    for _ in range(args.warmup):
        render_distributed(scene, cam, io_warm, seed=1, flags=flags)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rgb, mean = render_distributed(scene, cam, io, seed=1, flags=flags)
        if rank == 0:
            host_rgb.copy_(rgb, non_blocking=True)  # the frame's bytes on the host: where `pixels()` ends
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.rehearse_on_one_gpu else "cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    result = None
    if rank == 0:
        assert bool(host_rgb.any())
        samples = WIDTH * HEIGHT * spp * args.steps
        value = samples / elapsed / 1e6
        # Per-kernel timing: a profiled render (HIP events around every launch, on the launch stream) of this rank's share at the
        # sample count of one whole pass (64 M paths; 104 spp at 800x800 on one GPU), so that every launch of the profile is a
        # full-size one and a kernel trace of `bench.py --spp <that> --one-stream` averages over launches of exactly this kind.
        n_owned = max(1, WIDTH * HEIGHT // world)
        prof_spp = max(1, min(spp, (64 << 20) // n_owned))
        prof_io = rt.ImageOptions(HEIGHT, WIDTH, prof_spp, DEPTH)
        dmean = torch.empty((HEIGHT, WIDTH, 3), dtype=torch.float64, device="cuda")
        scene.render_device(cam, prof_io, rt.RenderOptions(seed=1, rank=0, world_size=world, tile_w=16, tile_h=16, flags=_abi.RT_RENDER_PROFILE), dmean.data_ptr())
        torch.cuda.synchronize()
        prof = scene.profile()
        roofline, cpu = None, None
        if not args.no_cpu_baseline:
            cpu, cnt = cpu_baseline_and_work(args.config, earth, timed=world == 1)
            trav, shade, raygen = algorithmic_flops(cnt)
            launches = max(prof["traverse_launches"], 1)
            seg_per_launch = prof["traverse_segments"] / launches
            common = {"profile_spp": prof_spp,
                      "flops_per_sample_all_kernels": round((trav + shade + raygen) / cnt["samples"], 1),
                      "segments_per_sample": round(cnt["segments"] / cnt["samples"], 2),
                      "kernel_ms": {k: round(prof[k + "_ms"], 2) for k in _abi.RT_K_NAMES}}
            fused = prof["traverse_launches"] == 0  # a handful of leaves: one kernel per bounce walks and shades (no hit records)
            if not fused and prof["traverse_ms"] >= prof["shade_ms"]:
                trav_per_seg = trav / cnt["segments"]
                ms_per_launch = prof["traverse_ms"] / launches
                achieved = trav_per_seg * seg_per_launch / (ms_per_launch * 1e-3) / 1e12
                pmc = pmc_counters(args.config, "k_traverse")
                roofline = {"bound": "valu", "kernel": "k_traverse", "achieved": round(achieved, 3), "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                            "frac": round(achieved / PEAK_FP32_TFLOPS, 4),
                            "traffic": round(pmc["hbm_bytes_per_segment"] * seg_per_launch) if pmc and "hbm_bytes_per_segment" in pmc else None,
                            "flops_per_segment": round(trav_per_seg, 1), "ms_per_launch": round(ms_per_launch, 4), "counters": pmc}
            else:
                # every path ends exactly once, all but a sliver of them in a shading launch: dead = paths, alive = segments - paths
                paths, segs = prof["samples"], prof["traverse_segments"]
                seg_per_launch = segs / max(prof["shade_launches"], 1)
                # (one kernel per bounce: no hit record, and the first bounce makes its own camera rays instead of reading them)
                bytes_in = (SHADE_BYTES_IN - 8) * max(segs - paths, 0) / max(segs, 1) if fused else SHADE_BYTES_IN
                bytes_per_seg = bytes_in + (SHADE_BYTES_OUT_ALIVE * max(segs - paths, 0) + SHADE_BYTES_OUT_DEAD * min(paths, segs)) / max(segs, 1)
                ms_per_launch = prof["shade_ms"] / max(prof["shade_launches"], 1)
                achieved = bytes_per_seg * seg_per_launch / (ms_per_launch * 1e-3) / 1e9
                pmc = pmc_counters(args.config, "k_shade")
                flops_per_seg = (trav + shade) / cnt["segments"] if fused else shade / cnt["segments"]
                roofline = {"bound": "hbm", "kernel": "k_shade<SHADE_SCAN> (walk + shade)" if fused else "k_shade", "achieved": round(achieved, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                            "frac": round(achieved / PEAK_HBM_GBPS, 4),
                            "traffic": round(pmc["hbm_bytes_per_segment"] * seg_per_launch) if pmc and "hbm_bytes_per_segment" in pmc else None,
                            "bytes_per_segment": round(bytes_per_seg, 1), "ms_per_launch": round(ms_per_launch, 4),
                            "valu_frac": round(flops_per_seg * seg_per_launch / (ms_per_launch * 1e-3) / 1e12 / PEAK_FP32_TFLOPS, 4), "counters": pmc}
            roofline.update(common)
            roofline["segments_per_launch"] = round(seg_per_launch)
        result = {"metric": f"Msamples/sec (whole node) on {args.config}", "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world,
                  "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
                  "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                  "config": {"workload": f"{args.config} {WIDTH}x{HEIGHT}, {spp} spp, depth {DEPTH}, interleaved 16x16 tiles over {world} GPU(s), gather to GPU0, "
                                         f"image on the host", "scene_seed": 1, "render_seed": 1, "objects": n_objects},
                  "roofline": roofline, "cpu_baseline": cpu}
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return result


if __name__ == "__main__":
    main()
