"""Per-kernel time per path segment for the five example scenes (wavefront kernels only, one stream, profiled)."""
import sys
sys.path.insert(0, '.')
import raytracing_amd as rt
earth = rt.earthmap()
for name, w, h, spp in [('random_spheres', 1200, 675, 64), ('cornell', 600, 600, 128), ('perlin', 1920, 1080, 32), ('cornell_smoke', 800, 800, 64), ('final_scene', 800, 800, 32)]:
    sc, co = rt.Scene.example(name, 1, earth if name == 'final_scene' else None)
    cam = rt.Camera(co)
    io = rt.ImageOptions(h, w, spp, 50)
    sc.render(cam, rt.ImageOptions(h, w, 2, 50), rt.RenderOptions(seed=1))
    sc.render(cam, io, rt.RenderOptions(seed=1, flags=1 | 2 | 16))
    p = sc.profile()
    seg = p['segments']
    print('%-15s samples %.1f M segments/sample %.2f  traverse %.3f ns/segment  shade %.3f ns/segment  (G segments/s: traverse %.1f, shade %.1f)' % (
        name, p['samples'] / 1e6, seg / p['samples'], 1e6 * p['traverse_ms'] / seg, 1e6 * p['shade_ms'] / seg, seg / p['traverse_ms'] / 1e6, seg / p['shade_ms'] / 1e6), flush=True)
