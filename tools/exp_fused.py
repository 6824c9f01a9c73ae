"""One kernel per bounce against two (RT_RENDER_ONE_KERNEL / RT_RENDER_TWO_KERNELS) on the four configurations: frame time and
identity of the frames.  argv: spp scale (default 0.25 of each configuration's sample count)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import torch
import raytracing_amd as rt
from raytracing_amd import _abi
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.25
CONFIGS = {"cornell": (600, 600, 1000), "perlin": (1920, 1080, 1000), "cornell_smoke": (800, 800, 4000), "final_scene": (800, 800, 10000)}
for name, (w, h, spp) in CONFIGS.items():
    if name == "final_scene" and len(sys.argv) < 3:
        continue
    spp = max(8, int(spp * scale))
    sc, co = rt.Scene.example(name, 1, rt.earthmap() if name == "final_scene" else None)
    cam = rt.Camera(co)
    buf = torch.empty((h, w, 3), dtype=torch.float64, device="cuda")
    frames = {}
    for label, flags in (("two kernels", _abi.RT_RENDER_TWO_KERNELS), ("one kernel", _abi.RT_RENDER_ONE_KERNEL), ("auto", 0)):
        for s in (min(64, spp), spp):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            sc.render_device(cam, rt.ImageOptions(h, w, s, 50), rt.RenderOptions(seed=1, flags=flags | _abi.RT_RENDER_RECALIBRATE), buf.data_ptr())
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        frames[label] = buf.cpu().numpy().copy()
        print(f"{name} {w}x{h} {spp} spp, {label}: {dt:.3f} s, {w * h * spp / dt / 1e6:.0f} Msamples/s, same frame as two kernels: {np.array_equal(frames[label], frames['two kernels'], equal_nan=True)}", flush=True)
