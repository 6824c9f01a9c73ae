"""Stress check on the GPU: for random image sizes, sample counts, depths, pass sizes, tile sizes and rank splits, every
render mode (default two-stream, one stream, wavefront only, tail only, nodes in global memory) must give the same bytes."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import raytracing_amd as rt
from raytracing_amd import _abi

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
earth = rt.earthmap()
scenes = {n: rt.Scene.example(n, 1, earth if n == 'final_scene' else None) for n in ('random_spheres', 'cornell', 'perlin', 'cornell_smoke', 'final_scene')}
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for case in range(n_cases):
    name = list(scenes)[case % 5]
    sc, co = scenes[name]
    cam = rt.Camera(co)
    w, h = int(rng.integers(2, 200)), int(rng.integers(2, 200))
    spp, depth = int(rng.integers(1, 40)), int(rng.choice([1, 2, 3, 7, 50]))
    io = rt.ImageOptions(h, w, spp, depth)
    seed = int(rng.integers(1, 1 << 40))
    _, base = sc.render(cam, io, rt.RenderOptions(seed=seed), want_mean=True)
    assert np.isfinite(base).all() or name in ('final_scene',), (name, w, h)
    variants = [rt.RenderOptions(seed=seed, flags=f, samples_per_pass=int(p), tile_w=int(tw), tile_h=int(th))
                for f, p, tw, th in [(_abi.RT_RENDER_ONE_STREAM, 0, 0, 0), (_abi.RT_RENDER_NO_TAIL, rng.integers(1, spp + 1), 0, 0),
                                     (_abi.RT_RENDER_TAIL_ONLY, 0, rng.integers(1, 33), rng.integers(1, 33)),
                                     (_abi.RT_RENDER_NODES_IN_GLOBAL, rng.integers(1, spp + 1), 0, 0), (0, rng.integers(1, spp + 1), 0, 0),
                                     (_abi.RT_RENDER_PROFILE, 0, 0, 0)]]
    for ro in variants:
        _, m = sc.render(cam, io, ro, want_mean=True)
        assert np.array_equal(m, base, equal_nan=True), (name, w, h, spp, depth, ro.flags, ro.samples_per_pass)
    world = int(rng.integers(2, 6))
    parts = [sc.render(cam, io, rt.RenderOptions(seed=seed, rank=r, world_size=world), want_mean=True)[1] for r in range(world)]
    assert np.array_equal(np.nansum(parts, axis=0), np.nan_to_num(base)), (name, world)
    print('case', case, name, w, h, spp, depth, 'ok', flush=True)
print('all invariant')
