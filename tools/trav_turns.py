"""Analysis: what the waves of k_traverse spend their turns on (needs librt_hip.so built with -DRT_TRAV_STATS).
final_scene 800x800 at 16 spp, wavefront kernels only, one stream."""
import ctypes as C
import sys
sys.path.insert(0, '.')
import raytracing_amd as rt
from raytracing_amd import _abi
lib = C.CDLL(_abi.HIP_LIB_PATH)
sc, co = rt.Scene.example('final_scene', 1, rt.earthmap()); cam = rt.Camera(co)
out = (C.c_ulonglong * 16)()
keep = 256 if len(sys.argv) > 1 and sys.argv[1] == 'keep-tree' else 0   # RT_RENDER_KEEP_TREE: the tree grouped by box areas
sc.render(cam, rt.ImageOptions(800, 800, 16, 50), rt.RenderOptions(seed=1, flags=keep))   # (lends rays: the tree is regrouped before the counted frame)
print('box tests per sampled ray: %.1f by areas, %.1f in use' % (sc.profile()['tree_tests_by_area'], sc.profile()['tree_tests_sampled']))
lib.rt_debug_trav_stats(out)
sc.render(cam, rt.ImageOptions(800, 800, 16, 50), rt.RenderOptions(seed=1, flags=1 | 2 | 16 | keep))
seg = sc.profile()['segments']
assert lib.rt_debug_trav_stats(out) == 0
s = list(out)
names = ['step turns', 'lanes able to step at a step turn', 'leaf turns', 'lanes with a leaf at a leaf turn', 'sphere leaves', 'box leaves', 'other leaves',
         'leaf turns with more than one kind', 'refills', 'rays taken', 'leaves tried in round 1 (spheres)', 'leaves tried in round 2 (the rest)',
         'leaf turns with a plain rectangle in round 2', 'rectangle lanes in round 2', 'leaf turns with another kind in round 2', 'lanes of other kinds in round 2']
for n, v in zip(names, s):
    print('%-40s %14d  per segment %.4f' % (n, v, v / seg))
print('segments', seg)
print('lanes able to step per step turn: %.1f of 64' % (s[1] / s[0]))
print('lanes with a leaf per leaf turn: %.1f (sphere %.1f, box %.1f, other %.2f); turns with more than one kind: %.1f %%' % (s[3] / s[2], s[4] / s[2], s[5] / s[2], s[6] / s[2], 100 * s[7] / s[2]))
print('rays per refill: %.1f' % (s[9] / s[8]))
print('leaves tried per leaf turn: %.1f' % ((s[10] + s[11]) / s[2]))
print('leaf turns with a rectangle: %.1f %% (%.1f lanes each); with a leaf that is neither sphere, box nor rectangle: %.1f %% (%.1f lanes each)' % (100 * s[12] / s[2], s[13] / max(s[12], 1), 100 * s[14] / s[2], s[15] / max(s[14], 1)))
