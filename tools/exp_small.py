"""Experiment: the four small example scenes with and without RT_RENDER_ONE_STREAM."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import raytracing_amd as rt
for name, w, h, s in [('cornell', 600, 600, 256), ('perlin', 1920, 1080, 64), ('cornell_smoke', 800, 800, 128), ('random_spheres', 400, 225, 256)]:
    sc2, co2 = rt.Scene.example(name, 1); cam2 = rt.Camera(co2)
    io2 = rt.ImageOptions(h, w, s, 50)
    sc2.render(cam2, rt.ImageOptions(h, w, 4, 50))
    for flags in (0, 16, 0, 16):
        t0 = time.time(); sc2.render(cam2, io2, rt.RenderOptions(seed=1, flags=flags)); t1 = time.time()
        print(name, 'flags', flags, '%.4fs Msamples/s %.1f' % (t1 - t0, w * h * s / (t1 - t0) / 1e6), flush=True)
