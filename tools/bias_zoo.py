"""The many-seed comparison of tools/bias_check.py on a scene the reference's examples do not contain: a Cornell room holding
the wrapped shapes and the materials side by side (boxes behind rotations about each axis, behind a translation off the
fp32 grid, a tilted two-sided panel, a glass box, a metal box, a medium in a rotated box, a moving sphere, a metal sphere).

    python tools/bias_zoo.py oracle <n> [first seed]   -> gpurun_out/zoo_oracle.npz     (CPU; copied to tests/golden/frame_means/zoo_room_oracle.npz with its metadata)
    python tools/bias_zoo.py gpu <n> [first seed]      -> gpurun_out/zoo_gpu.npz        (GPU box)
    python tools/bias_zoo.py compare <oracle npz> <gpu npz>
"""
import os
import sys
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import numpy as np
W, H, K, SPP, DEPTH = 96, 96, 8, 2048, 50


from scenes_util import zoo_room  # noqa: E402


def block_means(m):
    c = np.clip(np.nan_to_num(np.asarray(m, np.float64), nan=0.0), 0.0, 1.0)
    return c.reshape(H // K, K, W // K, K, 3).mean(axis=(1, 3))


def main():
    mode = sys.argv[1]
    if mode == "compare":
        from tools.bias_check import compare
        o, g = (np.asarray(np.load(p)["blocks"], np.float64) for p in sys.argv[2:4])
        diff, se, mean, z = compare(o, g)
        print(f"zoo room: {len(g)} gpu against {len(o)} oracle renders; frame mean gpu - oracle {np.round(diff, 6)} = {np.round(diff / se, 2)} standard errors, {np.round(100 * diff / mean, 3)} %")
        print(f"zoo room: block z-scores: mean square {np.mean(z ** 2):.3f}, largest |z| {np.abs(z).max():.2f}")
        np.set_printoptions(linewidth=200, precision=1, suppress=True)
        print(z.mean(axis=2))
        return
    import raytracing_amd as rt
    n = int(sys.argv[2])
    first = int(sys.argv[3]) if len(sys.argv) > 3 else (5001 if mode == "oracle" else 7001)
    objects, background, co_kw = zoo_room()
    co = rt.CameraOptions(**co_kw)
    blocks = []
    os.makedirs("gpurun_out", exist_ok=True)
    if mode == "oracle":
        from oracle_lib import OracleScene
        osc, oco = OracleScene(objects, background), co.to_c()
        for s in range(first, first + n):
            _, m, secs, _ = osc.render(oco, W, H, SPP, DEPTH, seed=s, threads=min(16, os.cpu_count() or 1))
            blocks.append(block_means(m))
            np.savez_compressed(os.path.join("gpurun_out", "zoo_oracle.npz"), blocks=np.array(blocks, np.float32))
            print("oracle seed", s, blocks[-1].mean(axis=(0, 1)), "%.0f s" % secs, flush=True)
    else:
        sc, cam = rt.Scene(objects, background), rt.Camera(co)
        for s in range(first, first + n):
            _, m = sc.render(cam, rt.ImageOptions(H, W, SPP, DEPTH), rt.RenderOptions(seed=s), want_mean=True)
            blocks.append(block_means(m))
        np.savez_compressed(os.path.join("gpurun_out", "zoo_gpu.npz"), blocks=np.array(blocks, np.float32))
        print("gpu", n, "seeds done", np.array(blocks).mean(axis=(0, 1, 2)))


if __name__ == "__main__":
    main()
