"""Is the GPU renderer's converged frame biased against the CPU oracle?  Many independent renders on each side of the small
frames of tests/golden/make_converged.py (4 096 spp, depth 50); kept per render: the means of the frame's blocks (per channel,
of the per-pixel means clamped to [0, 1]).  The two sides' averages are compared against the scatter between seeds, over the
frame and block by block.

    python tools/bias_check.py oracle <scene> <n> [first seed]    (CPU, here)   -> tests/golden/frame_means/<scene>_oracle.npz
    python tools/bias_check.py gpu <scene> <n> [first seed]       (GPU box)     -> gpurun_out/bias_<scene>_gpu.npz
    python tools/bias_check.py compare <scene> <gpu npz>
"""
import os
import sys
import time
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import numpy as np
FRAMES = {"cornell": (96, 96, 8), "perlin": (128, 72, 8), "cornell_smoke": (96, 96, 8), "final_scene": (96, 96, 8), "random_spheres": (96, 54, 6)}  # width, height, block
SPP, DEPTH, SCENE_SEED = 4096, 50, 1
FIXTURES = os.path.join("tests", "golden", "frame_means")


def block_means(m, k):
    c = np.clip(np.nan_to_num(np.asarray(m, np.float64), nan=0.0), 0.0, 1.0)
    h, w = c.shape[:2]
    assert h % k == 0 and w % k == 0
    return c.reshape(h // k, k, w // k, k, 3).mean(axis=(1, 3))


def compare(o, g):
    """o, g: (renders, by, bx, 3) block means of the two sides.  Returns frame-mean difference, its standard error, the blocks'
    z-scores (pooled variance of the two sides per block)."""
    fo, fg = o.mean(axis=(1, 2)), g.mean(axis=(1, 2))
    diff = fg.mean(0) - fo.mean(0)
    se = np.sqrt(fo.var(0, ddof=1) / len(fo) + fg.var(0, ddof=1) / len(fg))
    pooled = ((len(o) - 1) * o.var(0, ddof=1) + (len(g) - 1) * g.var(0, ddof=1)) / (len(o) + len(g) - 2)
    z = (g.mean(0) - o.mean(0)) / np.sqrt(np.maximum(pooled, 1e-300) * (1.0 / len(o) + 1.0 / len(g)))
    z = np.where(pooled > 0, z, 0.0)   # (blocks that are the same constant on both sides, e.g. saturated ones)
    return diff, se, fo.mean(0), z


def main():
    mode, name = sys.argv[1], sys.argv[2]
    w, h, k = FRAMES[name]
    if mode == "compare":
        o, g = (np.asarray(a, np.float64) for a in (np.load(os.path.join(FIXTURES, f"{name}_oracle.npz"))["blocks"], np.load(sys.argv[3])["blocks"]))
        diff, se, mean, z = compare(o, g)
        print(f"{name}: {len(g)} gpu against {len(o)} oracle renders; frame mean gpu - oracle {np.round(diff, 6)} = {np.round(diff / se, 2)} standard errors, {np.round(100 * diff / mean, 3)} %")
        print(f"{name}: block z-scores ({z.shape[0]}x{z.shape[1]} blocks x 3 channels): mean square {np.mean(z ** 2):.3f}, largest |z| {np.abs(z).max():.2f}")
        return
    import raytracing_amd as rt
    earth = rt.earthmap() if name == "final_scene" else None
    n, first = int(sys.argv[3]), int(sys.argv[4]) if len(sys.argv) > 4 else (5001 if mode == "oracle" else 7001)
    if mode == "oracle":
        from oracle_lib import OracleScene
        path = os.path.join(FIXTURES, f"{name}_oracle.npz")
        os.makedirs(FIXTURES, exist_ok=True)
        seeds, blocks = [], []
        if os.path.exists(path):
            fx = np.load(path)
            seeds, blocks = list(fx["seeds"]), list(fx["blocks"])
        osc, oco = OracleScene.example(name, SCENE_SEED, earth)
        for s in range(first, first + n):
            if s in seeds:
                continue
            t0 = time.time()
            _, m, _, _ = osc.render(oco, w, h, SPP, DEPTH, seed=s, threads=min(16, os.cpu_count() or 1))
            seeds.append(s); blocks.append(block_means(m, k))
            np.savez_compressed(path, seeds=np.array(seeds), blocks=np.array(blocks, np.float32), width=w, height=h, block=k, spp=SPP, depth=DEPTH, scene_seed=SCENE_SEED)
            print(name, "oracle seed", s, blocks[-1].mean(axis=(0, 1)), "%.0f s" % (time.time() - t0), flush=True)
    else:
        sc, co = rt.Scene.example(name, SCENE_SEED, earth)
        blocks = []
        for s in range(first, first + n):
            _, m = sc.render(rt.Camera(co), rt.ImageOptions(h, w, SPP, DEPTH), rt.RenderOptions(seed=s), want_mean=True)
            blocks.append(block_means(m, k))
        os.makedirs("gpurun_out", exist_ok=True)
        np.savez_compressed(os.path.join("gpurun_out", f"bias_{name}_gpu.npz"), blocks=np.array(blocks, np.float32))
        print(name, "gpu", n, "seeds done")


if __name__ == "__main__":
    main()
