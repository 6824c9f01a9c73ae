"""Hit-record parity at a larger sample than the unit tests: camera rays and five generations of scattered rays of every
example scene, GPU (rt_intersect_batch) against the oracle; prints mismatch counts and the largest relative errors."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import raytracing_amd as rt
from oracle_lib import OracleScene
from scenes_util import EXAMPLES, path_rays

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
earth = rt.earthmap()
for name in EXAMPLES:
    e = earth if name == 'final_scene' else None
    sc, co = rt.Scene.example(name, 1, e)
    osc, oco = OracleScene.example(name, 1, e)
    tot = flag = prim = 0
    worst = {}
    for o, d, t in path_rays(osc, oco, 200, 150, 5, n, 5, np.random.default_rng(17)):
        ref = osc.intersect(o, d, t, seed=9)
        got = sc.intersect_batch(o, d, t, seed=9)
        tot += len(ref)
        fm = got['hit'] != ref['hit']; flag += int(fm.sum())
        both = (got['hit'] == 1) & (ref['hit'] == 1)
        pm = both & (got['prim'] != ref['prim']); prim += int(pm.sum())
        ok = both & ~pm
        scale = np.maximum(1.0, np.maximum(np.abs(ref['point'][ok]).max(axis=1), np.abs(o[ok]).max(axis=1)))
        for f in ('t', 'normal', 'in_vec', 'u', 'v', 'point'):
            a, b = got[f][ok].astype(np.float64), ref[f][ok]
            if a.ndim == 1:
                err = np.abs(a - b) / np.maximum(1.0, np.abs(b))
            elif f == 'point':
                err = (np.abs(a - b) / scale[:, None]).max(axis=1)
            else:
                err = (np.abs(a - b) / np.maximum(1.0, np.abs(b))).max(axis=1)
            err = err[np.isfinite(err)]
            if len(err):
                worst[f] = max(worst.get(f, 0.0), float(err.max()))
    print(f"{name}: {tot} rays, hit/miss mismatches {flag}, primitive mismatches {prim}, worst relative errors " +
          ", ".join(f"{k} {v:.1e}" for k, v in worst.items()), flush=True)
