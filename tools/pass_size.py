"""final_scene 800x800: frame rate against the number of samples a pass holds (workspace 120 bytes per path)."""
import sys, time
sys.path.insert(0, '.')
import raytracing_amd as rt
sc, co = rt.Scene.example('final_scene', 1, rt.earthmap()); cam = rt.Camera(co)
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 1664
for S in (52, 104, 208, 416):
    sc.render(cam, rt.ImageOptions(800, 800, S, 50), rt.RenderOptions(seed=1, samples_per_pass=S))   # sizes the workspace, calibrates
    t0 = time.time(); sc.render(cam, rt.ImageOptions(800, 800, spp, 50), rt.RenderOptions(seed=1, samples_per_pass=S)); t1 = time.time()
    print('samples per pass', S, 'paths per pass %.0f M' % (S * 0.64), '%.3f s' % (t1 - t0), '%.1f Msamples/s' % (640000 * spp / (t1 - t0) / 1e6), flush=True)
