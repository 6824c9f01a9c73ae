"""Benchmark of the path-tracing hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config NAME]

N > 1 needs one process per GPU.  Under a launcher (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`:
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment) this file is one rank.  Called plainly with --gpus N > 1 it
starts that launcher itself, before anything touches a GPU, passes rank 0's JSON line through and exits with the launcher's
code.

One step = one full frame of a BASELINE configuration, tile-split over the N ranks, gathered on rank 0 (one RCCL
sum-reduce of the f64 frame), quantised there and copied to the host: the metric runs to image-on-host.  The default
configuration is the one BASELINE.json quotes its metric on: the reference's final_scene example at 800x800, 10 000
samples per pixel, 50 reflections.  --config picks one of the other GPU configurations (cornell 600x600 at 1 000 spp,
perlin 1920x1080 at 1 000 spp, cornell_smoke 800x800 at 4 000 spp).  The scene is resident on the device before the
timed region; warm-up steps render the same frame at the sample count of one full pass of the rank's share plus a
little (they size the workspace and let the library learn the schedule of passes of exactly the size the timed frame runs).

Rank 0 prints one JSON line.  `ranks` lists, per rank, the device it rendered on (index, name, PCI bus id) and the number of
pixels it owned, gathered over the process group: N distinct bus ids are N distinct GPUs.  `roofline` prices the kernel that
takes most of the frame.  For the BVH traversal that is the fp32 vector peak: algorithmic flops per path segment are counted
by the CPU oracle on the reference's traversal order (per-operation constants below).  For shading it is the HBM peak: 56
bytes read per segment (ray, throughput, hit record) and 48 written when the path goes on or 16 when it ends.  The scenes of
a handful of leaves run a bounce as one kernel that walks and shades; it sits under neither roof (it waits on its own
dependent loads), so its line says "latency" and carries both fractions.  Either is divided by the kernel's average launch
duration, measured here with HIP events in a profiled render after the timed one (kernels one after the other on one
stream, which is the duration a kernel trace of `bench.py --one-stream` shows; the timed render overlaps the traversal of
one half of a pass with the shading of the other half on two streams where that pays).  `roofline.counters` are the
rocprofv3 PMC figures of the same kernel on the same scene (VALU busy, L2 hit rate, HBM bytes per segment), read from the
newest profiles/*_pmc.json; that file names the sha256 of the kernel sources it was collected on, and when the sources
here differ the counters are marked stale and `traffic` (that file's bytes per segment times this run's segments per
launch) is left out.  `cpu_baseline` is the CPU oracle (an f64 restatement of the reference, not the Rust binary) timed on
rank 0's host, after the timed region and while the other ranks sleep, on a bounded sample of the same frame; it states
the host's CPU model and core counts and how many threads it used.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

DEPTH = 50
# name: (width, height, samples per pixel, objects) -- the four GPU configurations of BASELINE.json
CONFIGS = {"cornell": (600, 600, 1000, 8), "perlin": (1920, 1080, 1000, 4), "cornell_smoke": (800, 800, 4000, 8), "final_scene": (800, 800, 10000, 1410)}
PEAK_FP32_TFLOPS = 157.3  # MI355X fp32 vector peak
PEAK_HBM_GBPS = 8000.0    # MI355X HBM3E peak
KERNEL_SOURCES = ("raytracing_amd/csrc/rt_hip.hip", "raytracing_amd/csrc/rt_core.h")

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


def kernel_source_sha256():
    """sha256 over the kernel sources the library in this tree is built from (what tools/pmc_summarise.py writes beside the counters)."""
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def pmc_counters(config, kernel):
    """The rocprofv3 PMC figures of `kernel` on `config` from profiles/*_pmc.json (made by tools/pmc_summarise.py), or
    None.  `stale`: the file was collected on other kernel sources than the ones here (or does not say)."""
    d = os.path.join(ROOT, "profiles")
    files = sorted(f for f in os.listdir(d) if f.endswith("_pmc.json")) if os.path.isdir(d) else []
    if not files:
        return None
    # the collection made on the kernel sources that are here, if there is one; else the last by name (stale)
    docs = []
    for f in files:
        with open(os.path.join(d, f)) as fh:
            docs.append((f, json.load(fh)))
    sha = kernel_source_sha256()
    match = [fd for fd in docs if fd[1].get("kernel_source_sha256") == sha]
    name, doc = (match or docs)[-1]
    files = [name]
    rec = doc.get("configs", {}).get(config, {}).get("kernels", {}).get(kernel)
    if rec is None:
        return None
    out = {k: rec[k] for k in ("valu_busy", "wait_share", "l2_hit_rate", "hbm_bytes_per_segment", "hbm_gbps", "valu_lane_slots_per_segment") if k in rec}
    out["source"] = "profiles/" + files[-1]
    out["stale"] = doc.get("kernel_source_sha256") != sha
    return out


def host_cpus():
    """What the host offers and what this process may use of it: logical CPUs, the CPUs of this process's affinity mask, the
    cgroup's CPU quota, OMP_NUM_THREADS (the share a GPU box grants one job), the model string of /proc/cpuinfo."""
    info = {"logical_cpus": os.cpu_count() or 1, "cpu_model": None, "affinity_cpus": None, "cgroup_cpus": None, "omp_num_threads": None}
    try:
        info["affinity_cpus"] = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.lower().startswith("model name"):
                    info["cpu_model"] = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()[:2]
            if quota != "max":
                info["cgroup_cpus"] = max(1, int(quota) // int(period))
    except (OSError, ValueError):
        pass
    try:
        info["omp_num_threads"] = int(os.environ["OMP_NUM_THREADS"])
    except (KeyError, ValueError):
        pass
    limits = [v for v in (info["logical_cpus"], info["affinity_cpus"], info["cgroup_cpus"], info["omp_num_threads"]) if v]
    info["usable_cpus"] = max(1, min(limits))
    return info


def cpu_baseline_and_work(name, earth, timed, threads=0):
    """Counts the reference algorithm's work per sample with the oracle (a 96-wide frame, 8 spp) and, if `timed`, times the
    oracle on a bounded sample of the configuration's frame: the full view at 2/5 of the size, the sample count chosen from a
    pilot render so that the timed render takes about 15 s.  One thread per CPU this process may use (every logical CPU of the
    host unless an affinity mask, a cgroup quota or OMP_NUM_THREADS grants less)."""
    from oracle_lib import OracleScene
    w, h, _, _ = CONFIGS[name]
    cpus = host_cpus()
    threads = threads or cpus["usable_cpus"]
    osc, oco = OracleScene.example(name, 1, earth)
    cw = 96 if w == h else 128
    ch = cw * h // w
    _, _, _, cnt = osc.render(oco, cw, ch, 8, DEPTH, seed=1, threads=threads, counters=True)
    if not timed:
        return None, cnt
    sw, sh = w * 2 // 5, h * 2 // 5
    _, _, pilot, _ = osc.render(oco, sw, sh, 4, DEPTH, seed=1, threads=threads, shade_every=True)
    spp = int(max(8, min(4096, round(15.0 / max(pilot, 1e-3) * 4))))
    _, _, secs, _ = osc.render(oco, sw, sh, spp, DEPTH, seed=1, threads=threads, shade_every=True)
    value = sw * sh * spp / secs / 1e6
    base = {"value": round(value, 4), "unit": "Msamples/s", "cores": threads, "kind": "port",
            "per_core": round(value / threads, 5), "host_logical_cpus": cpus["logical_cpus"], "host_usable_cpus": cpus["usable_cpus"],
            "cpu_model": cpus["cpu_model"], "limits": {k: cpus[k] for k in ("affinity_cpus", "cgroup_cpus", "omp_num_threads")},
            "sample": f"{name} {sw}x{sh}, {spp} spp, depth {DEPTH}: {sw * sh * spp} samples in {secs:.1f} s; f64 C++ restatement of the reference "
                      f"(reference traversal order, every accepted candidate shaded), {threads} std::threads on {cpus['cpu_model']} "
                      f"({cpus['logical_cpus']} logical CPUs on the host, {cpus['usable_cpus']} usable by this job)"}
    return base, cnt


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="final_scene", help="BASELINE configuration to run (default: the one the metric is quoted on)")
    ap.add_argument("--spp", type=int, default=0, help="samples per pixel of a timed step (default: the configuration's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (default: every CPU this process may use)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true", help="N > 1 ranks that all use GPU 0 and gather over gloo (functional rehearsal only)")
    ap.add_argument("--one-stream", action="store_true", help="render every pass as one half on one stream (RT_RENDER_ONE_STREAM): kernels do not "
                    "overlap, so a kernel trace shows the per-kernel durations that the roofline is quoted on")
    return ap.parse_args(argv)


def launch_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: N fresh rank processes through torch.distributed.run, started before this
    process has touched a GPU; their output passes through (rank 0 prints the JSON line); returns the launcher's exit code."""
    import torch  # (counting devices does not initialise them)
    have = torch.cuda.device_count()
    if not args.rehearse_on_one_gpu and have < args.gpus:
        print(f"bench.py: --gpus {args.gpus} but {have} GPU(s) are visible", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd)


def device_identity(torch, index):
    """Name and PCI bus id of HIP device `index` (the bus id from the runtime: torch's properties carry it only in some builds)."""
    props = torch.cuda.get_device_properties(index)
    bus = None
    try:
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        buf = ctypes.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, int(index)) == 0:
            bus = buf.value.decode()
    except OSError:
        pass
    if bus is None and hasattr(props, "pci_bus_id"):
        bus = f"{getattr(props, 'pci_domain_id', 0):04x}:{props.pci_bus_id:02x}:{getattr(props, 'pci_device_id', 0):02x}.0"
    return props.name, bus


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        sys.exit(launch_ranks(args, argv))

    import datetime
    import torch
    import torch.distributed as dist
    import raytracing_amd as rt
    from raytracing_amd import _abi
    from raytracing_amd.distributed import TILE_H, TILE_W, owned_mask, render_distributed

    WIDTH, HEIGHT, SPP, n_objects = CONFIGS[args.config]
    spp = args.spp or SPP
    PASS_PATHS = int(_abi.hip_lib().rt_auto_pass_paths())  # paths of a full pass (the library's automatic pass size)
    under_launcher = "WORLD_SIZE" in os.environ
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    if args.rehearse_on_one_gpu:
        local_rank = 0  # every rank on GPU 0, frames gathered over gloo: exercises the N > 1 path on a one-GPU box (not a measurement)
    torch.cuda.set_device(local_rank)
    if under_launcher:  # (also for one rank under a launcher: the collectives then run on a group of one)
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo", timeout=datetime.timedelta(minutes=30))
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(minutes=30))
    grouped = dist.is_initialized()

    earth = rt.earthmap() if args.config == "final_scene" else None
    scene, co = rt.Scene.example(args.config, 1, earth)
    cam = rt.Camera(co)
    scene.device(local_rank)

    # who renders what, gathered over the group itself
    name, bus = device_identity(torch, local_rank)
    me = {"rank": rank, "device": local_rank, "name": name, "pci_bus_id": bus, "host": socket.gethostname(), "pid": os.getpid(),
          "n_owned": int(owned_mask(WIDTH, HEIGHT, rank, world, TILE_W, TILE_H).sum())}
    ranks = [me]
    if grouped:
        ranks = [None] * world
        dist.all_gather_object(ranks, me)
    # warm-up frames: one full pass of the smallest share and a little more, so that the library has learnt the schedule of
    # passes of the size the timed frame is made of
    pass_spp = max(1, PASS_PATHS // max(1, min(r["n_owned"] for r in ranks)))
    io = rt.ImageOptions(HEIGHT, WIDTH, spp, DEPTH)
    io_warm = rt.ImageOptions(HEIGHT, WIDTH, min(spp, pass_spp + 24), DEPTH)

    def sync():
        torch.cuda.synchronize()
        if grouped:
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
    schedule = scene.profile()
    if grouped:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.rehearse_on_one_gpu else "cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # Everything below is rank 0's; the other ranks sleep on the group's store meanwhile (a barrier would have them spin on
    # the CPUs the oracle is timed on).
    store = dist.distributed_c10d._get_default_store() if grouped and world > 1 else None
    result = None
    if rank == 0:
        assert bool(host_rgb.any())
        samples = WIDTH * HEIGHT * spp * args.steps
        value = samples / elapsed / 1e6
        # Per-kernel timing: a profiled render (HIP events around every launch, on the launch stream) of this rank's share at the
        # sample count of one whole pass (128 M paths; 209 spp at 800x800 on one GPU), so that every launch of the profile is a
        # full-size one and a kernel trace of `bench.py --spp <that> --one-stream` averages over launches of exactly this kind.
        prof_spp = max(1, min(spp, PASS_PATHS // max(1, me["n_owned"])))
        prof_io = rt.ImageOptions(HEIGHT, WIDTH, prof_spp, DEPTH)
        dmean = torch.empty((HEIGHT, WIDTH, 3), dtype=torch.float64, device="cuda")
        scene.render_device(cam, prof_io, rt.RenderOptions(seed=1, rank=0, world_size=world, tile_w=TILE_W, tile_h=TILE_H, flags=_abi.RT_RENDER_PROFILE), dmean.data_ptr())
        torch.cuda.synchronize()
        prof = scene.profile()
        roofline, cpu = None, None
        if not args.no_cpu_baseline:
            cpu, cnt = cpu_baseline_and_work(args.config, earth, timed=True, threads=args.cpu_threads)
            trav, shade, raygen = algorithmic_flops(cnt)
            launches = max(prof["traverse_launches"], 1)
            seg_per_launch = prof["traverse_segments"] / launches
            common = {"profile_spp": prof_spp,
                      "flops_per_sample_all_kernels": round((trav + shade + raygen) / cnt["samples"], 1),
                      "segments_per_sample": round(cnt["segments"] / cnt["samples"], 2),
                      "kernel_ms": {k: round(prof[k + "_ms"], 2) for k in _abi.RT_K_NAMES}}
            fused = prof["traverse_launches"] == 0  # a handful of leaves: one kernel per bounce walks and shades (no hit records)

            def traffic_of(pmc):
                return round(pmc["hbm_bytes_per_segment"] * seg_per_launch) if pmc and not pmc["stale"] and "hbm_bytes_per_segment" in pmc else None

            if not fused and prof["traverse_ms"] >= prof["shade_ms"]:
                trav_per_seg = trav / cnt["segments"]
                ms_per_launch = prof["traverse_ms"] / launches
                achieved = trav_per_seg * seg_per_launch / (ms_per_launch * 1e-3) / 1e12
                pmc = pmc_counters(args.config, "k_traverse")
                roofline = {"bound": "valu", "kernel": "k_traverse", "achieved": round(achieved, 3), "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                            "frac": round(achieved / PEAK_FP32_TFLOPS, 4), "traffic": traffic_of(pmc),
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
                hbm_frac = achieved / PEAK_HBM_GBPS
                valu_frac = flops_per_seg * seg_per_launch / (ms_per_launch * 1e-3) / 1e12 / PEAK_FP32_TFLOPS
                roofline = {"bound": "latency" if fused else "hbm", "kernel": "k_shade<SHADE_SCAN> (walk + shade)" if fused else "k_shade",
                            "achieved": round(achieved, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": round(hbm_frac, 4),
                            "hbm_frac": round(hbm_frac, 4), "valu_frac": round(valu_frac, 4), "traffic": traffic_of(pmc),
                            "bytes_per_segment": round(bytes_per_seg, 1), "flops_per_segment": round(flops_per_seg, 1), "ms_per_launch": round(ms_per_launch, 4),
                            "counters": pmc}
                if fused:
                    # (the profiled launches are one bounce each; in the timed frame the first launch of a pass may run two, see schedule)
                    roofline["note"] = ("under neither roof: the kernel's waves wait on their own dependent loads (counters.wait_share) with the vector "
                                        "units counters.valu_busy busy; frac is the HBM fraction, valu_frac the fp32 vector fraction of the same launches")
            roofline.update(common)
            roofline["segments_per_launch"] = round(seg_per_launch)
        result = {"metric": f"Msamples/sec (whole node) on {args.config}", "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world,
                  "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
                  "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                  "config": {"workload": f"{args.config} {WIDTH}x{HEIGHT}, {spp} spp, depth {DEPTH}, interleaved {TILE_W}x{TILE_H} tiles over {world} GPU(s), gather to GPU0, "
                                         f"image on the host", "scene_seed": 1, "render_seed": 1, "objects": n_objects, "warmup_spp": io_warm.samples_per_pixel,
                             "backend": (dist.get_backend() if grouped else None)},
                  "ranks": ranks, "distinct_gpus": len({(r["host"], r["pci_bus_id"]) for r in ranks}),
                  "schedule": {"pass_paths": schedule["pass_paths"], "tail_from": schedule["tail_from"], "two_stream_passes": schedule["two_stream_passes"],
                               "passes": schedule["passes"], "two_bounce_passes": schedule["two_bounce_passes"]},
                  "roofline": roofline, "cpu_baseline": cpu}
        print(json.dumps(result), flush=True)
        if store is not None:
            store.set("bench/rank0_done", "1")
    elif store is not None:
        store.wait(["bench/rank0_done"], datetime.timedelta(minutes=30))
    if grouped:
        dist.barrier()
        dist.destroy_process_group()
    return result


if __name__ == "__main__":
    main()
