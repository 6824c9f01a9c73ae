import sys, time
sys.path.insert(0, '.')
import raytracing_amd as rt
sc, co = rt.Scene.example('final_scene', 1, rt.earthmap()); cam = rt.Camera(co)
sc.render(cam, rt.ImageOptions(800, 800, 16, 50), rt.RenderOptions(seed=1))
for spp_pass in (0, 52, 104, 156, 208, 312):
    io = rt.ImageOptions(800, 800, 1248, 50)
    t0 = time.time(); sc.render(cam, io, rt.RenderOptions(seed=1, samples_per_pass=spp_pass)); t1 = time.time()
    print('samples per pass', spp_pass, '%.3f s' % (t1 - t0), 'Msamples/s %.1f' % (800 * 800 * 1248 / (t1 - t0) / 1e6), flush=True)
