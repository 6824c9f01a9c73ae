"""Analysis: is the GPU renderer's converged frame biased against the CPU oracle?  Frame means (per channel, of the per-pixel
means clamped to [0, 1]) of many independent renders on each side; the difference of the two sides' averages is judged against
the scatter between seeds.  The frames are those of tests/golden/make_converged.py (small frame, 4 096 spp, depth 50).

    python tools/bias_check.py oracle <scene> <n> [first seed]    (CPU, here)   -> tests/golden/frame_means/<scene>_oracle.json
    python tools/bias_check.py gpu <scene> <n> [first seed]       (GPU box)     -> gpurun_out/bias_<scene>_gpu.json
    python tools/bias_check.py compare <scene> <gpu json>
"""
import json
import os
import sys
import time
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import numpy as np
FRAMES = {"cornell": (96, 96), "perlin": (128, 72), "cornell_smoke": (96, 96), "final_scene": (96, 96), "random_spheres": (96, 54)}
SPP, DEPTH, SCENE_SEED = 4096, 50, 1


def frame_mean(m):
    return np.clip(np.nan_to_num(np.asarray(m, np.float64), nan=0.0), 0.0, 1.0).reshape(-1, 3).mean(0).tolist()


def main():
    mode, name = sys.argv[1], sys.argv[2]
    w, h = FRAMES[name]
    import raytracing_amd as rt
    earth = rt.earthmap() if name == "final_scene" else None
    if mode == "oracle":
        from oracle_lib import OracleScene
        n, first = int(sys.argv[3]), int(sys.argv[4]) if len(sys.argv) > 4 else 5001
        path = os.path.join("tests", "golden", "frame_means", f"{name}_oracle.json")
        os.makedirs(os.path.dirname(path), exist_ok=True)
        doc = json.load(open(path)) if os.path.exists(path) else {"scene": name, "width": w, "height": h, "spp": SPP, "depth": DEPTH, "means": {}}
        osc, oco = OracleScene.example(name, SCENE_SEED, earth)
        for s in range(first, first + n):
            if str(s) in doc["means"]:
                continue
            t0 = time.time()
            _, m, _, _ = osc.render(oco, w, h, SPP, DEPTH, seed=s, threads=min(16, os.cpu_count() or 1))
            doc["means"][str(s)] = frame_mean(m)
            json.dump(doc, open(path, "w"), indent=1)
            print(name, "oracle seed", s, doc["means"][str(s)], "%.0f s" % (time.time() - t0), flush=True)
    elif mode == "gpu":
        n, first = int(sys.argv[3]), int(sys.argv[4]) if len(sys.argv) > 4 else 7001
        sc, co = rt.Scene.example(name, SCENE_SEED, earth)
        doc = {"scene": name, "means": {}}
        for s in range(first, first + n):
            _, m = sc.render(rt.Camera(co), rt.ImageOptions(h, w, SPP, DEPTH), rt.RenderOptions(seed=s), want_mean=True)
            doc["means"][str(s)] = frame_mean(m)
        os.makedirs("gpurun_out", exist_ok=True)
        json.dump(doc, open(os.path.join("gpurun_out", f"bias_{name}_gpu.json"), "w"), indent=1)
        print(name, "gpu", n, "seeds done")
    else:
        o = np.array(list(json.load(open(os.path.join("tests", "golden", "frame_means", f"{name}_oracle.json")))["means"].values()))
        g = np.array(list(json.load(open(sys.argv[3]))["means"].values()))
        se = np.sqrt(o.var(0, ddof=1) / len(o) + g.var(0, ddof=1) / len(g))
        print(f"{name}: oracle {len(o)} renders, mean {np.round(o.mean(0), 6)}, seed-to-seed std {np.round(o.std(0, ddof=1), 6)}")
        print(f"{name}: gpu    {len(g)} renders, mean {np.round(g.mean(0), 6)}, seed-to-seed std {np.round(g.std(0, ddof=1), 6)}")
        print(f"{name}: gpu - oracle {np.round(g.mean(0) - o.mean(0), 6)} = {np.round((g.mean(0) - o.mean(0)) / se, 2)} standard errors; relative {np.round((g.mean(0) - o.mean(0)) / o.mean(0) * 100, 3)} %")


if __name__ == "__main__":
    main()
