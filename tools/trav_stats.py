"""Traversal work per path segment on the CPU build of the kernel code: node tests, leaves reached and primitive tests
for the rays of a few hundred camera paths (final_scene unless another example is named)."""
import ctypes as C
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import raytracing_amd as rt
from raytracing_amd import _abi
from oracle_lib import OracleScene, TWIN_PATH, rays_to_c
from scenes_util import path_rays

name = sys.argv[1] if len(sys.argv) > 1 else 'final_scene'
earth = rt.earthmap() if name == 'final_scene' else None
sc, co = rt.Scene.example(name, 1, earth)
osc, oco = OracleScene.example(name, 1, earth)
lib = C.CDLL(TWIN_PATH)
lib.twin_traverse_stats.argtypes = [C.POINTER(_abi.rt_scene_desc), C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32]
desc = sc._host.rth_scene_desc(sc._h)
otot = 0
tot = np.zeros(3); n = 0; dtot = {k: 0 for k in (1, 2, 3, 4, 8)}
lib.twin_traverse_deferred.argtypes = [C.POINTER(_abi.rt_scene_desc), C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p]
for o, d, t in path_rays(osc, oco, 64, 64, 5, 3000, 12, np.random.default_rng(3)):
    rays = rays_to_c(o, d, t)
    out = np.zeros((len(rays), 3), dtype=np.uint32)
    lib.twin_traverse_stats(desc, rays.ctypes.data, len(rays), 1, out.ctypes.data, None, 0)
    tot += out.sum(axis=0); n += len(rays)
    o3 = np.zeros(len(rays), dtype=np.uint32)
    lib.twin_traverse_ordered.argtypes = [C.POINTER(_abi.rt_scene_desc), C.c_void_p, C.c_uint64, C.c_void_p]
    lib.twin_traverse_ordered(desc, rays.ctypes.data, len(rays), o3.ctypes.data); otot += int(o3.sum())
    for k in dtot:
        o2 = np.zeros(len(rays), dtype=np.uint32)
        lib.twin_traverse_deferred(desc, rays.ctypes.data, len(rays), k, o2.ctypes.data)
        dtot[k] += int(o2.sum())
lib.twin_generic_leaves.restype = C.c_ulonglong
print('general primitive tests per segment (stats + ordered + deferred runs together count each ray %d times): %.3f' % (2 + len(dtot), lib.twin_generic_leaves() / n / (2 + len(dtot))))
print(name, 'segments', n, 'nodes/seg %.2f leaves/seg %.2f prim tests/seg %.2f' % tuple(tot / n))
lib.twin_traverse_ordered.argtypes = [C.POINTER(_abi.rt_scene_desc), C.c_void_p, C.c_uint64, C.c_void_p]
print('node tests/seg with k leaves pending:', {k: round(v / n, 2) for k, v in dtot.items()})
print('node tests/seg, reference binary tree walked nearer child first (not result-neutral): %.2f' % (otot / n))
