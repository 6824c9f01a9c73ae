"""Minimal workload for rocprofv3: final_scene 800x800 at a few samples per pixel.
argv: spp (default 16), render flags (default 2 = RT_RENDER_NO_TAIL: wavefront kernels only, one stream)."""
import sys
sys.path.insert(0, '.')
import raytracing_amd as rt
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
flags = int(sys.argv[2]) if len(sys.argv) > 2 else 2
sc, co = rt.Scene.example('final_scene', 1, rt.earthmap())
cam = rt.Camera(co)
sc.render(cam, rt.ImageOptions(800, 800, spp, 50), rt.RenderOptions(seed=1, flags=flags))
