"""Analysis: where in the frame the GPU differs from the oracle's fixture frames.  argv: scene, n seeds.  Writes the mean of n GPU
renders (clamped linear means) to gpurun_out/biasmap_<scene>.npy together with the count of NaN pixels per render."""
import os
import sys
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import numpy as np
import raytracing_amd as rt
name, n = sys.argv[1], int(sys.argv[2])
fx = np.load(os.path.join("tests", "golden", f"converged_{name}.npz"))
w, h, spp, depth = (int(fx[k]) for k in ("width", "height", "spp", "depth"))
sc, co = rt.Scene.example(name, int(fx["scene_seed"]), rt.earthmap() if name == "final_scene" else None)
acc = np.zeros((h, w, 3)); nans = []
for s in range(9001, 9001 + n):
    _, m = sc.render(rt.Camera(co), rt.ImageOptions(h, w, spp, depth), rt.RenderOptions(seed=s), want_mean=True)
    nans.append(int(np.isnan(m).any(axis=2).sum()))
    acc += np.clip(np.nan_to_num(m, nan=0.0), 0.0, 1.0)
np.save(os.path.join("gpurun_out", f"biasmap_{name}.npy"), acc / n)
print(name, "NaN pixels per render:", nans, "oracle A, B:", int(np.isnan(fx["mean_a"]).any(axis=2).sum()), int(np.isnan(fx["mean_b"]).any(axis=2).sum()))
