"""Writes tests/golden/oracle_hits.json: hit records and small-render statistics produced by the
CPU oracle (oracle/rt_oracle.cpp) for the five example scenes.  The reference itself cannot be
run here (Rust), so these pin the oracle against regressions, not against the reference."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import raytracing_amd as rt  # noqa: E402
from oracle_lib import OracleScene, oracle_camera_rays  # noqa: E402

out = {}
rng = np.random.default_rng(2024)
earth = rt.earthmap()
for name in ["random_spheres", "cornell", "perlin", "cornell_smoke", "final_scene"]:
    sc, co = OracleScene.example(name, 1, earth if name == "final_scene" else None)
    w, h, spp, n = 12, 12, 4, 48
    o, d, t = oracle_camera_rays(co, w, h, 11, rng.integers(0, w, n), rng.integers(0, h, n), rng.integers(0, 9, n))
    o, d, t = o.astype(np.float32), d.astype(np.float32), t.astype(np.float32)
    hits = sc.intersect(o, d, t, seed=11)
    rgb, mean, _, _ = sc.render(co, w, h, spp, 50, seed=1, threads=4)
    out[name] = {"seed": 11, "w": w, "h": h, "spp": spp, "o": o.tolist(), "d": d.tolist(), "time": t.tolist(), "hit": [int(x) for x in hits["hit"]],
                 "prim": [int(x) for x in hits["prim"]], "t": hits["t"].tolist(), "normal": hits["normal"].tolist(),
                 "mean_rgb": mean.mean(axis=(0, 1)).tolist(), "rgb_sum": int(rgb.astype(np.uint64).sum())}
with open(os.path.join(HERE, "oracle_hits.json"), "w") as f:
    json.dump(out, f)
print("wrote", os.path.join(HERE, "oracle_hits.json"))
