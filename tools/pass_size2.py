"""final_scene 800x800: frame rate against the number of samples per pass (the library's automatic choice is 64 M paths a pass)."""
import sys, time
sys.path.insert(0, '.')
import raytracing_amd as rt
sc, co = rt.Scene.example('final_scene', 1, rt.earthmap()); cam = rt.Camera(co)
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 2496
for per_pass in [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0,52,104,208,416").split(',')]:
    ro = rt.RenderOptions(seed=1, samples_per_pass=per_pass)
    sc.render(cam, rt.ImageOptions(800, 800, max(per_pass, 104) + 8, 50), ro)      # sizes the workspace, learns the schedule of this pass size
    t0 = time.time(); sc.render(cam, rt.ImageOptions(800, 800, spp, 50), ro); t1 = time.time()
    p = sc.profile()
    print('samples per pass', per_pass, 'spp', spp, '%.3f s' % (t1 - t0), 'Msamples/s %.1f' % (640000 * spp / (t1 - t0) / 1e6), 'passes', p['passes'], 'two-stream', p['two_stream_passes'], 'tail from', p['tail_from'], flush=True)
