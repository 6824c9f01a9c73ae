"""Analysis: what the waves of k_shade hold on final_scene (needs librt_hip.so built with -DRT_SHADE_STATS).
800x800 at 16 spp, wavefront kernels only, one stream."""
import ctypes as C
import sys
sys.path.insert(0, '.')
import raytracing_amd as rt
from raytracing_amd import _abi
lib = C.CDLL(_abi.HIP_LIB_PATH)
name = sys.argv[1] if len(sys.argv) > 1 else 'final_scene'
sc, co = rt.Scene.example(name, 1, rt.earthmap() if name == 'final_scene' else None); cam = rt.Camera(co)
out = (C.c_ulonglong * 40)()
sc.render(cam, rt.ImageOptions(800, 800, 16, 50), rt.RenderOptions(seed=1, flags=128))
lib.rt_debug_shade_stats(out)
sc.render(cam, rt.ImageOptions(800, 800, 16, 50), rt.RenderOptions(seed=1, flags=1 | 2 | 16 | 128))
assert lib.rt_debug_shade_stats(out) == 0
s = list(out)
names = ['all', 'hit', 'miss', 'sphere', 'box', 'rectangle', 'medium', 'behind wrappers', 'surface coordinates wanted', 'Lambertian, solid colour inlined',
         'Lambertian, texture call', 'metal', 'dielectric', 'light', 'isotropic', 'checker texture', 'noise texture', 'image texture', 'moving sphere']
print('segments', sc.profile()['segments'])
for k, n in enumerate(names):
    print('%-36s waves %10d (%.3f of all)  lanes %12d (%.1f per such wave, %.4f of live lanes)' % (n, s[2 * k], s[2 * k] / max(s[0], 1), s[2 * k + 1], s[2 * k + 1] / max(s[2 * k], 1), s[2 * k + 1] / max(s[1], 1)))
