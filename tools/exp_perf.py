import sys, time, os, shutil
sys.path.insert(0, '.')
lib = sys.argv[1]
import raytracing_amd._abi as _abi
if lib != 'base':
    _abi.HIP_LIB_PATH = os.path.join(os.path.dirname(_abi.HIP_LIB_PATH), lib)
import raytracing_amd as rt
sc, co = rt.Scene.example('final_scene', 1, rt.earthmap()); cam = rt.Camera(co)
sc.render(cam, rt.ImageOptions(800, 800, 16, 50), rt.RenderOptions(seed=1))
sc.render(cam, rt.ImageOptions(800, 800, 64, 50), rt.RenderOptions(seed=1, flags=1)); p = sc.profile()
print(lib, 'traverse_ms', round(p['traverse_ms'], 1), 'shade_ms', round(p['shade_ms'], 1), flush=True)
