"""One scene, one frame after a warm-up, for counter runs: python tools/pmc_one.py <scene> <w> <h> <spp>"""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import raytracing_amd as rt
name, w, h, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
sc, co = rt.Scene.example(name, 1, rt.earthmap() if name == 'final_scene' else None); cam = rt.Camera(co)
sc.render(cam, rt.ImageOptions(h, w, spp, 50), rt.RenderOptions(seed=1, flags=rt._abi.RT_RENDER_ONE_STREAM))
