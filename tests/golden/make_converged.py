"""Makes the converged-image fixtures of tests/golden/converged_<scene>.npz.

    python tests/golden/make_converged.py [scene ...]

For each of the four GPU configurations' scenes (cornell, perlin, cornell_smoke, final_scene) the CPU oracle
(oracle/rt_oracle.cpp, the f64 restatement of the reference) renders a small frame at 4 096 samples per pixel, 50
reflections, twice with independent render seeds A = 1001 and B = 2002.  Stored per scene: the two frames of linear f64
mean radiance (row 0 = top), the frame size, spp, depth, seeds and scene seed.  The GPU test renders the same frame with
a third seed and compares with A; the distance between A and B is the noise floor the comparison is measured against.
Runs in the build container (8 threads, a few minutes per scene); nothing here touches the GPU or the reference.
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

SPP, DEPTH, SEED_A, SEED_B, SCENE_SEED = 4096, 50, 1001, 2002, 1
FRAMES = {"cornell": (192, 192), "perlin": (256, 144), "cornell_smoke": (192, 192), "final_scene": (192, 192)}   # (width, height)


def main(names):
    import raytracing_amd as rt
    from oracle_lib import OracleScene
    threads = min(16, os.cpu_count() or 1)
    for name in names:
        w, h = FRAMES[name]
        earth = rt.earthmap() if name == "final_scene" else None
        osc, oco = OracleScene.example(name, SCENE_SEED, earth)
        t0 = time.time()
        _, a, _, _ = osc.render(oco, w, h, SPP, DEPTH, seed=SEED_A, threads=threads)
        _, b, _, _ = osc.render(oco, w, h, SPP, DEPTH, seed=SEED_B, threads=threads)
        out = os.path.join(os.environ.get("CONVERGED_OUT", HERE), f"converged_{name}.npz")
        # (f32 is plenty: the frames are compared in 8-bit units, and the noise floor at 4 096 spp is 1e-2 of a pixel value)
        np.savez_compressed(out, mean_a=a.astype(np.float32), mean_b=b.astype(np.float32), width=w, height=h, spp=SPP, depth=DEPTH, seed_a=SEED_A, seed_b=SEED_B, scene_seed=SCENE_SEED)
        print(f"{name}: {w}x{h} at {SPP} spp twice in {time.time() - t0:.0f} s -> {out} ({os.path.getsize(out) // 1024} KB)", flush=True)


if __name__ == "__main__":
    main(sys.argv[1:] or list(FRAMES))
