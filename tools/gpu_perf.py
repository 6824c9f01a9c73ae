"""Quick performance probe: final_scene 800x800 at a few sample counts, with kernel timings."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import raytracing_amd as rt
from raytracing_amd import _abi
earth = rt.earthmap()
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 128
flags_list = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else [0]
sc, co = rt.Scene.example('final_scene', 1, earth); cam = rt.Camera(co)
io = rt.ImageOptions(800, 800, spp, 50)
sc.render(cam, rt.ImageOptions(800, 800, 16, 50), rt.RenderOptions(seed=1))
ref = None
for flags in flags_list:
    t0 = time.time(); rgb, mean = sc.render(cam, io, rt.RenderOptions(seed=1, flags=flags), want_mean=True); t1 = time.time()
    _ = sc.render(cam, io, rt.RenderOptions(seed=1, flags=flags | 1)); p = sc.profile()
    if ref is None: ref = mean
    print('flags', flags, 'spp', spp, '%.3fs' % (t1 - t0), 'Msamples/s %.1f' % (800 * 800 * spp / (t1 - t0) / 1e6), 'identical to first:', np.array_equal(ref, mean, equal_nan=True), 'nan px', int(np.isnan(mean).any(axis=2).sum()),
          {k: round(v, 1) for k, v in p.items() if k.endswith('_ms')}, 'trav launches', p['traverse_launches'],
          'box tests per sampled ray %.1f -> %.1f' % (p['tree_tests_by_area'], p['tree_tests_sampled']), flush=True)
for name, w, h, s in [('cornell', 600, 600, 256), ('perlin', 1920, 1080, 64), ('cornell_smoke', 800, 800, 128), ('random_spheres', 400, 225, 256)]:
    sc2, co2 = rt.Scene.example(name, 1); cam2 = rt.Camera(co2)
    io2 = rt.ImageOptions(h, w, s, 50)
    sc2.render(cam2, rt.ImageOptions(h, w, 4, 50))
    t0 = time.time(); sc2.render(cam2, io2, rt.RenderOptions(seed=1)); t1 = time.time()
    print(name, w, h, s, '%.3fs Msamples/s %.1f' % (t1 - t0, w * h * s / (t1 - t0) / 1e6), flush=True)
