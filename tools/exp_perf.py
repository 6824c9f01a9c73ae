"""Experiment: kernel times of final_scene 800x800 at 16 spp, wavefront kernels only, one stream."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import raytracing_amd as rt
sc, co = rt.Scene.example('final_scene', 1, rt.earthmap()); cam = rt.Camera(co)
io = rt.ImageOptions(800, 800, 16, 50)
sc.render(cam, io, rt.RenderOptions(seed=1))
sc.render(cam, io, rt.RenderOptions(seed=1, flags=1 | 2))
p = sc.profile()
print({k: round(v, 2) for k, v in p.items() if k.endswith('_ms')})
