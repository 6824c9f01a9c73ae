"""Analysis: the frame-mean check of tests/test_gpu_converged.py for several GPU seeds, to tell a bias from the luck of one seed.
argv: scene [seeds...].  Prints (mean(gpu) - mean(A)) / sigma and (mean(gpu) - mean(B)) / sigma per channel, signed."""
import os
import sys
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import numpy as np
import raytracing_amd as rt
name = sys.argv[1] if len(sys.argv) > 1 else "cornell_smoke"
seeds = [int(x) for x in sys.argv[2:]] or [3003, 1, 2, 3, 4, 5, 6, 7]
fx = np.load(os.path.join("tests", "golden", f"converged_{name}.npz"))
w, h, spp, depth = (int(fx[k]) for k in ("width", "height", "spp", "depth"))
sc, co = rt.Scene.example(name, int(fx["scene_seed"]), rt.earthmap() if name == "final_scene" else None)
clip = lambda m: np.clip(np.nan_to_num(m, nan=0.0), 0.0, 1.0)
ca, cb = clip(fx["mean_a"]), clip(fx["mean_b"])
n = ca.shape[0] * ca.shape[1]
sigma = (ca - cb).reshape(-1, 3).std(axis=0) / np.sqrt(n)
print("A - B over sigma", np.round((ca.reshape(-1, 3).mean(0) - cb.reshape(-1, 3).mean(0)) / sigma, 2))
tot = []
for s in seeds:
    _, mean = sc.render(rt.Camera(co), rt.ImageOptions(h, w, spp, depth), rt.RenderOptions(seed=s), want_mean=True)
    cg = clip(mean)
    da = (cg.reshape(-1, 3).mean(0) - ca.reshape(-1, 3).mean(0)) / sigma
    db = (cg.reshape(-1, 3).mean(0) - cb.reshape(-1, 3).mean(0)) / sigma
    tot.append((da + db) / 2)
    print("seed", s, "vs A", np.round(da, 2), "vs B", np.round(db, 2))
tot = np.array(tot)
print("gpu - (A + B) / 2 over sigma: mean over seeds", np.round(tot.mean(0), 2), "std", np.round(tot.std(0), 2), "(expected 0 +- %.2f, std 0.71)" % (np.sqrt(0.75 / len(seeds))))
