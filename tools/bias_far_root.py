"""The many-seed comparison of tools/bias_check.py on tests/scenes_util.far_root_variant_scene() (spheres that are also hit from
within and materials that see the facing normal: switches that are not the reference's behaviour; the oracle implements them).
    python tools/bias_far_root.py oracle|gpu <n>  -> gpurun_out/far_root_<side>.npz;  python tools/bias_far_root.py compare"""
import os
import sys
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import numpy as np
W, H, K, SPP, DEPTH = 96, 64, 8, 2048, 50
from scenes_util import far_root_variant_scene  # noqa: E402
import raytracing_amd as rt  # noqa: E402


def block_means(m):
    c = np.clip(np.nan_to_num(np.asarray(m, np.float64), nan=0.0), 0.0, 1.0)
    return c.reshape(H // K, K, W // K, K, 3).mean(axis=(1, 3))


mode = sys.argv[1]
if mode == "compare":
    from tools.bias_check import compare
    o, g = (np.asarray(np.load(os.path.join("gpurun_out", f"far_root_{s}.npz"))["blocks"], np.float64) for s in ("oracle", "gpu"))
    diff, se, mean, z = compare(o, g)
    print(f"far-root scene: {len(g)} gpu against {len(o)} oracle renders; frame mean gpu - oracle {np.round(diff, 6)} = {np.round(diff / se, 2)} standard errors, {np.round(100 * diff / mean, 3)} %")
    print(f"far-root scene: block z-scores: mean square {np.mean(z ** 2):.3f}, largest |z| {np.abs(z).max():.2f}")
    sys.exit(0)
n = int(sys.argv[2])
objects, background, co_kw = far_root_variant_scene()
co = rt.CameraOptions(**co_kw)
blocks = []
os.makedirs("gpurun_out", exist_ok=True)
if mode == "oracle":
    from oracle_lib import OracleScene
    osc, oco = OracleScene(objects, background), co.to_c()
    for s in range(5001, 5001 + n):
        _, m, secs, _ = osc.render(oco, W, H, SPP, DEPTH, seed=s, threads=min(16, os.cpu_count() or 1))
        blocks.append(block_means(m))
        print("oracle seed", s, blocks[-1].mean(axis=(0, 1)), "%.0f s" % secs, flush=True)
else:
    sc, cam = rt.Scene(objects, background), rt.Camera(co)
    for s in range(7001, 7001 + n):
        _, m = sc.render(cam, rt.ImageOptions(H, W, SPP, DEPTH), rt.RenderOptions(seed=s), want_mean=True)
        blocks.append(block_means(m))
np.savez_compressed(os.path.join("gpurun_out", f"far_root_{mode}.npz"), blocks=np.array(blocks, np.float32))
print(mode, n, "renders done")
