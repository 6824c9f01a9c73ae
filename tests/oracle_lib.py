"""ctypes access to the test-only libraries: the f64 oracle (oracle/liboracle.so) and the
CPU build of the kernel arithmetic (tests/twin/librt_twin.so).  Used by tests/, smoke() and
bench.py's cpu_baseline only."""
import ctypes as C
import os
import subprocess

import numpy as np

from raytracing_amd import _abi
from raytracing_amd.api import Backend, HIT_DTYPE, rays_to_c

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_PATH = os.path.join(ROOT, "oracle", "liboracle.so")
TWIN_PATH = os.path.join(ROOT, "tests", "twin", "librt_twin.so")

ORC_HIT_DTYPE = np.dtype([("hit", np.int32), ("prim", np.uint32), ("t", np.float64), ("point", np.float64, 3), ("normal", np.float64, 3),
                          ("in_vec", np.float64, 3), ("u", np.float64), ("v", np.float64)])

_P, _D = C.c_void_p, C.POINTER(C.c_double)


def build_test_libs():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "twin")], check=True)


_oracle = None
_twin = None


def oracle():
    global _oracle
    if _oracle is None:
        if not os.path.exists(ORACLE_PATH):
            build_test_libs()
        lib = C.CDLL(ORACLE_PATH)
        syms = dict(_abi.builder_symbols("orc_"))
        syms.update({
            "orc_scene_new": (_P, [C.c_size_t, C.POINTER(_P), C.POINTER(_P), _D, C.c_uint64]),
            "orc_example_scene": (_P, [C.c_char_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(_abi.camera_options)]),
            "orc_scene_num_objects": (C.c_size_t, [_P]),
            "orc_scene_bvh_stats": (None, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
            "orc_scene_leaf_order": (None, [_P, _P]),
            "orc_shape_intersect": (None, [_P, C.c_size_t, _P, _P, _P, C.c_double, C.c_double, _P, _P]),
            "orc_scene_intersect": (None, [_P, C.c_size_t, _P, _P, _P, C.c_double, C.c_double, C.c_uint64, _P, C.c_int, _P]),
            "orc_camera_rays": (None, [C.POINTER(_abi.camera_options), C.c_uint32, C.c_uint32, C.c_uint64, C.c_size_t, _P, _P, _P, _P, _P, _P]),
            "orc_camera_frame": (None, [C.POINTER(_abi.camera_options), _P]),
            "orc_material_scatter": (None, [_P, _P, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_int32), _P, _P, _P]),
            "orc_scene_scatter": (None, [_P, _P, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_int32), _P, _P, _P]),
            "orc_texture_color": (None, [_P, C.c_double, C.c_double, _D, _D]),
            "orc_render": (C.c_double, [_P, C.POINTER(_abi.camera_options), C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_int,
                                        C.c_uint32, C.c_uint32, _P, _P, _P]),
            "orc_counters_len": (C.c_size_t, []),
            "orc_counters_names": (C.c_char_p, []),
            "orc_perlin3": (C.c_double, [_P, C.c_double, C.c_double, C.c_double]),
            "orc_philox": (None, [_P, _P, _P]),
            "orc_random_unit_vector": (None, [C.c_double, C.c_double, _D]),
            "orc_inside_unit_disc": (None, [C.c_uint64, C.c_uint32, C.c_uint32, _D]),
            "orc_schlick": (C.c_double, [C.c_double, C.c_double]),
            "orc_quantise": (C.c_uint8, [C.c_double]),
            "orc_rotate": (None, [_D, C.c_int, C.c_double, _D]),
            "orc_ray_at": (None, [_D, _D, C.c_double, _D]),
            "orc_bbox_hit": (C.c_int, [_D, _D, _D, _D, C.c_double, C.c_double]),
        })
        _oracle = _abi.bind(lib, syms)
    return _oracle


def twin():
    global _twin
    if _twin is None:
        if not os.path.exists(TWIN_PATH):
            build_test_libs()
        lib = C.CDLL(TWIN_PATH)
        syms = {
            "twin_intersect_batch": (None, [C.POINTER(_abi.rt_scene_desc), C.c_int64, _P, C.c_uint64, C.c_float, C.c_float, C.c_uint64, _P, _P]),
            "twin_camera_rays": (None, [C.POINTER(_abi.rt_camera), C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, _P, _P, _P, _P]),
            "twin_material_scatter": (None, [C.POINTER(_abi.rt_scene_desc), _P, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_int32), _P, _P, _P]),
            "twin_perlin3": (C.c_float, [_P, C.c_float, C.c_float, C.c_float]),
            "twin_random_unit_vector": (None, [C.c_float, C.c_float, _P]),
            "twin_philox": (None, [_P, _P, _P]),
            "twin_quantise": (C.c_uint8, [C.c_double]),
            "twin_walk_nodes": (C.c_uint32, [C.POINTER(_abi.rt_scene_desc), _P, C.c_uint32]),
            "twin_set_sample_rays": (None, [_P, C.c_uint64]),
            "twin_side_of_surface": (None, [_P, _P, C.c_int, C.c_uint64, _P]),
            "twin_traverse_stats": (None, [C.POINTER(_abi.rt_scene_desc), _P, C.c_uint64, C.c_uint64, _P, _P, C.c_uint32]),
            "twin_render": (None, [C.POINTER(_abi.rt_scene_desc), C.POINTER(_abi.rt_camera), C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, _P]),
        }
        _twin = _abi.bind(lib, syms)
    return _twin


def oracle_backend():
    return Backend(oracle(), "orc_")


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a if shape is None else a.reshape(shape)


def _as_f64_rays(origins, directions, times):
    """Rays are defined by their fp32 values (what the GPU sees), widened exactly to f64 for the oracle."""
    r = rays_to_c(origins, directions, times)
    return _f64(r[:, 0:3]), _f64(r[:, 3:6]), _f64(r[:, 6])


class OracleShape:
    def __init__(self, shape):
        self.lib = oracle()
        self.h = shape.build(oracle_backend())

    def intersect(self, origins, directions, times, tmin, tmax, medium_u=None):
        o, d, t = _as_f64_rays(origins, directions, times)
        n = o.shape[0]
        out = np.zeros(n, dtype=ORC_HIT_DTYPE)
        mu = None if medium_u is None else _f64(np.broadcast_to(np.asarray(medium_u, dtype=np.float32), (n,)))
        self.lib.orc_shape_intersect(self.h, n, o.ctypes.data, d.ctypes.data, t.ctypes.data, tmin, tmax, mu.ctypes.data if mu is not None else None,
                                     out.ctypes.data)
        return out

    def bound(self):
        mn, mx = (C.c_double * 3)(), (C.c_double * 3)()
        self.lib.orc_shape_bound(self.h, mn, mx)
        return np.array(mn), np.array(mx)

    def __del__(self):
        try:
            self.lib.orc_shape_free(self.h)
        except Exception:
            pass


class OracleScene:
    def __init__(self, objects=None, background=(0, 0, 0), bvh_seed=1, _handle=None):
        self.lib = oracle()
        if _handle is not None:
            self.h = _handle
            return
        be = oracle_backend()
        shapes = [o.shape.build(be) for o in objects]
        mats = [o.material.build(be) for o in objects]
        n = len(objects)
        self.h = self.lib.orc_scene_new(n, (C.c_void_p * max(n, 1))(*shapes), (C.c_void_p * max(n, 1))(*mats), (C.c_double * 3)(*map(float, background)),
                                        bvh_seed)
        for s in shapes:
            be.shape_free(s)
        for m in mats:
            be.mat_free(m)
        if not self.h:
            raise ValueError("empty scene")

    @staticmethod
    def example(name, scene_seed=1, earthmap=None):
        lib = oracle()
        co = _abi.camera_options()
        if earthmap is not None:
            earthmap = np.ascontiguousarray(earthmap, dtype=np.uint8)
            h = lib.orc_example_scene(name.encode(), scene_seed, earthmap.shape[1], earthmap.shape[0], earthmap.ctypes.data, C.byref(co))
        else:
            h = lib.orc_example_scene(name.encode(), scene_seed, 0, 0, None, C.byref(co))
        if not h:
            raise ValueError("unknown example or missing earth map: " + name)
        return OracleScene(_handle=h), co

    def num_objects(self):
        return self.lib.orc_scene_num_objects(self.h)

    def leaf_order(self):
        out = np.zeros(self.num_objects(), dtype=np.uint32)
        self.lib.orc_scene_leaf_order(self.h, out.ctypes.data)
        return out

    def bvh_stats(self):
        n, d = C.c_uint64(), C.c_uint64()
        self.lib.orc_scene_bvh_stats(self.h, C.byref(n), C.byref(d))
        return n.value, d.value

    def intersect(self, origins, directions, times, tmin=0.001, tmax=float("inf"), seed=0, medium_u=None, brute=False, mode=None):
        o, d, t = _as_f64_rays(origins, directions, times)
        n = o.shape[0]
        out = np.zeros(n, dtype=ORC_HIT_DTYPE)
        mu = None if medium_u is None else _f64(np.broadcast_to(np.asarray(medium_u, dtype=np.float32), (n,)))
        self.lib.orc_scene_intersect(self.h, n, o.ctypes.data, d.ctypes.data, t.ctypes.data, tmin, tmax, seed, mu.ctypes.data if mu is not None else None,
                                     mode if mode is not None else (1 if brute else 0), out.ctypes.data)
        return out

    def render(self, cam_options_c, width, height, spp, depth, seed=1, threads=8, shade_every=False, rows=None, counters=False):
        """Returns (rgb u8 (h,w,3), mean f64 (h,w,3), seconds, counters dict or None)."""
        rgb = np.zeros((height, width, 3), dtype=np.uint8)
        mean = np.zeros((height, width, 3), dtype=np.float64)
        r0, r1 = rows if rows else (0, height)
        ncnt = self.lib.orc_counters_len()
        cnt = np.zeros(ncnt, dtype=np.uint64)
        secs = self.lib.orc_render(self.h, C.byref(cam_options_c), width, height, spp, depth, seed, threads, int(shade_every), r0, r1, rgb.ctypes.data,
                                   mean.ctypes.data, cnt.ctypes.data if counters else None)
        names = self.lib.orc_counters_names().decode().split(",")
        return rgb, mean, secs, (dict(zip(names, map(int, cnt))) if counters else None)

    def __del__(self):
        try:
            self.lib.orc_scene_free(self.h)
        except Exception:
            pass


def oracle_camera_rays(cam_options_c, width, height, seed, px_i, px_j, sample):
    lib = oracle()
    px_i, px_j, sample = (np.ascontiguousarray(a, dtype=np.uint32) for a in (px_i, px_j, sample))
    n = px_i.shape[0]
    o, d, t = np.zeros((n, 3)), np.zeros((n, 3)), np.zeros(n)
    lib.orc_camera_rays(C.byref(cam_options_c), width, height, seed, n, px_i.ctypes.data, px_j.ctypes.data, sample.ctypes.data, o.ctypes.data, d.ctypes.data,
                        t.ctypes.data)
    return o, d, t


def twin_intersect(scene, origins, directions, times, tmin=0.001, tmax=float("inf"), scope=-1, seed=0, medium_u=None):
    """CPU build of the kernel's intersect path on a host-library scene (raytracing_amd.Scene)."""
    rays = rays_to_c(origins, directions, times)
    n = rays.shape[0]
    out = np.zeros(n, dtype=HIT_DTYPE)
    mu = None if medium_u is None else np.ascontiguousarray(np.broadcast_to(np.asarray(medium_u, dtype=np.float32), (n,)))
    desc = scene._host.rth_scene_desc(scene._h)
    twin().twin_intersect_batch(desc, scope, rays.ctypes.data, n, tmin, tmax, seed, mu.ctypes.data if mu is not None else None, out.ctypes.data)
    return out


def twin_render(scene, camera, width, height, spp, depth, seed=1, threads=8):
    mean = np.zeros((height, width, 3), dtype=np.float64)
    desc = scene._host.rth_scene_desc(scene._h)
    twin().twin_render(desc, C.byref(camera.abi), width, height, spp, depth, seed, threads, mean.ctypes.data)
    return mean


def quantise_image(mean):
    """lib.rs:68-72 on an array of means (oracle's quantiser, vectorised)."""
    m = np.nan_to_num(np.asarray(mean, dtype=np.float64), nan=0.0)
    y = np.sqrt(np.clip(m, 0.0, 1.0)) * 255.999
    return np.minimum(np.floor(y), 255).astype(np.uint8)


def rmse(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)))
