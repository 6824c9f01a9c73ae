"""ctypes declarations of include/rt_abi.h and loaders for the two product libraries.

The libraries are built in-tree (raytracing_amd/csrc/Makefile).  There is no CPU
fallback: if a library is missing, loading raises and every caller fails.
"""
import ctypes as C
import os

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
HIP_LIB_PATH = os.path.join(_CSRC, "librt_hip.so")
HOST_LIB_PATH = os.path.join(_CSRC, "libraytracing_host.so")

RT_OK = 0
RT_ERR_INVALID_ARGUMENT, RT_ERR_EMPTY_SCENE, RT_ERR_UNSUPPORTED, RT_ERR_NO_DEVICE, RT_ERR_HIP, RT_ERR_OUT_OF_MEMORY = 1, 2, 3, 4, 5, 6
RT_RENDER_PROFILE, RT_RENDER_NO_TAIL, RT_RENDER_TAIL_ONLY, RT_RENDER_NODES_IN_GLOBAL, RT_RENDER_ONE_STREAM, RT_RENDER_RECALIBRATE, RT_RENDER_ONE_KERNEL, RT_RENDER_TWO_KERNELS = 1, 2, 4, 8, 16, 32, 64, 128
RT_RENDER_KEEP_TREE = 256
RT_GEOM_SPHERE, RT_GEOM_MOVING_SPHERE, RT_GEOM_RECT, RT_GEOM_BOX = 0, 1, 2, 3
RT_PRIM_FLIP, RT_PRIM_MEDIUM, RT_PRIM_INERT, RT_PRIM_BOTH_ROOTS, RT_PRIM_MULTI_SCATTER = 1, 2, 4, 8, 16
RT_MATF_FACE_NORMAL = 1
RT_BVH_NONE, RT_BVH_EXACT = 0xFFFFFFFF, 0x80000000
RT_NOISE_PERLIN, RT_NOISE_FBM, RT_NOISE_TURBULENCE = 0, 1, 2
RT_K_NAMES = ("raygen", "traverse", "shade", "tail", "resolve", "finalize")


class rt_prim(C.Structure):
    _fields_ = [("geom", C.c_uint8), ("flags", C.c_uint8), ("xf_count", C.c_uint16), ("xf_first", C.c_uint32), ("geom_index", C.c_uint32),
                ("material", C.c_uint32), ("density", C.c_float), ("xf_outer", C.c_uint32), ("reserved", C.c_uint32 * 2)]


class rt_sphere(C.Structure):
    _fields_ = [("center", C.c_double * 3), ("radius", C.c_double), ("motion", C.c_double * 3), ("start_time", C.c_double)]


class rt_rect(C.Structure):
    _fields_ = [("axis", C.c_uint32), ("reserved", C.c_uint32), ("x0", C.c_double), ("y0", C.c_double), ("x1", C.c_double), ("y1", C.c_double),
                ("height", C.c_double)]


class rt_box(C.Structure):
    _fields_ = [("min", C.c_double * 3), ("max", C.c_double * 3)]


class rt_xform(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("reserved", C.c_uint32), ("a", C.c_double), ("b", C.c_double), ("c", C.c_double)]


class rt_material(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("texture", C.c_uint32), ("albedo", C.c_float * 3), ("param", C.c_float), ("reserved", C.c_uint32 * 2)]


class rt_texture(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("color", C.c_float * 3), ("odd", C.c_uint32), ("even", C.c_uint32), ("scale", C.c_float), ("resource", C.c_uint32)]


class rt_image(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("texel_offset", C.c_uint64)]


class rt_bvh_node(C.Structure):
    _fields_ = [("min", C.c_float * 3), ("skip", C.c_uint32), ("max", C.c_float * 3), ("prim", C.c_uint32)]


class rt_scene_desc(C.Structure):
    _fields_ = [("abi_version", C.c_uint32), ("n_prims", C.c_uint32), ("n_spheres", C.c_uint32), ("n_rects", C.c_uint32), ("n_boxes", C.c_uint32),
                ("n_xforms", C.c_uint32), ("n_materials", C.c_uint32), ("n_textures", C.c_uint32), ("n_images", C.c_uint32),
                ("n_perm_tables", C.c_uint32), ("n_bvh_nodes", C.c_uint32), ("reserved0", C.c_uint32),
                ("prims", C.POINTER(rt_prim)), ("spheres", C.POINTER(rt_sphere)), ("rects", C.POINTER(rt_rect)), ("boxes", C.POINTER(rt_box)),
                ("xforms", C.POINTER(rt_xform)), ("materials", C.POINTER(rt_material)), ("textures", C.POINTER(rt_texture)),
                ("images", C.POINTER(rt_image)), ("texels", C.POINTER(C.c_uint8)), ("texel_bytes", C.c_uint64),
                ("perm_tables", C.POINTER(C.c_uint8)), ("bvh", C.POINTER(rt_bvh_node)), ("background", C.c_double * 3)]


class rt_camera(C.Structure):
    _fields_ = [("origin", C.c_double * 3), ("lower_left_corner", C.c_double * 3), ("horizontal", C.c_double * 3), ("vertical", C.c_double * 3),
                ("u", C.c_double * 3), ("v", C.c_double * 3), ("lens_radius", C.c_double), ("shutter_open", C.c_double), ("shutter_close", C.c_double)]


class rt_image_options(C.Structure):
    _fields_ = [("height", C.c_uint32), ("width", C.c_uint32), ("samples_per_pixel", C.c_uint64), ("max_reflections", C.c_uint64)]


class rt_render_options(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("rank", C.c_uint32), ("world_size", C.c_uint32), ("tile_w", C.c_uint32), ("tile_h", C.c_uint32),
                ("samples_per_pass", C.c_uint32), ("flags", C.c_uint32), ("stream", C.c_void_p)]


class rt_ray(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("direction", C.c_float * 3), ("time", C.c_float)]


class rt_hit(C.Structure):
    _fields_ = [("hit", C.c_int32), ("prim", C.c_uint32), ("t", C.c_float), ("point", C.c_float * 3), ("normal", C.c_float * 3),
                ("in_vec", C.c_float * 3), ("u", C.c_float), ("v", C.c_float)]


class rt_scatter(C.Structure):
    _fields_ = [("scattered", C.c_int32), ("color", C.c_float * 3), ("origin", C.c_float * 3), ("direction", C.c_float * 3)]


class rt_profile(C.Structure):
    _fields_ = [("kernel_ms", C.c_double * 6), ("kernel_launches", C.c_uint64 * 6), ("samples", C.c_uint64), ("segments", C.c_uint64),
                ("traverse_segments", C.c_uint64), ("passes", C.c_uint64), ("tree_tests_by_area", C.c_double), ("tree_tests_sampled", C.c_double),
                ("tail_from", C.c_int32), ("two_stream_passes", C.c_uint32), ("pass_paths", C.c_uint64),
                ("two_bounce_passes", C.c_uint32), ("reserved", C.c_uint32)]


class camera_options(C.Structure):
    """rth_camera_options / orc_camera_options: CameraOptions of src/lib.rs:405-425."""
    _fields_ = [("aperture", C.c_double), ("aspect_ratio", C.c_double), ("focus_distance", C.c_double), ("looking_at", C.c_double * 3),
                ("origin", C.c_double * 3), ("shutter_close", C.c_double), ("shutter_open", C.c_double), ("vfov", C.c_double), ("vup", C.c_double * 3)]


# Every symbol include/rt_abi.h declares: (restype, argtypes)
HIP_SYMBOLS = {
    "rt_scene_create": (C.c_int, [C.POINTER(rt_scene_desc), C.c_int, C.POINTER(C.c_void_p)]),
    "rt_scene_destroy": (None, [C.c_void_p]),
    "rt_render": (C.c_int, [C.c_void_p, C.POINTER(rt_camera), C.POINTER(rt_image_options), C.POINTER(rt_render_options), C.c_void_p, C.c_void_p]),
    "rt_render_device": (C.c_int, [C.c_void_p, C.POINTER(rt_camera), C.POINTER(rt_image_options), C.POINTER(rt_render_options), C.c_void_p]),
    "rt_render_device_noalloc": (C.c_int, [C.c_void_p, C.POINTER(rt_camera), C.POINTER(rt_image_options), C.POINTER(rt_render_options), C.c_void_p]),
    "rt_quantise_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
    "rt_intersect_batch": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(rt_ray), C.c_uint64, C.c_float, C.c_float, C.c_uint64, C.c_void_p, C.POINTER(rt_hit)]),
    "rt_scatter_batch": (C.c_int, [C.c_void_p, C.POINTER(rt_hit), C.c_uint64, C.c_uint64, C.c_void_p, C.POINTER(rt_scatter)]),
    "rt_camera_rays": (C.c_int, [C.c_void_p, C.POINTER(rt_camera), C.POINTER(rt_image_options), C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.POINTER(rt_ray)]),
    "rt_get_profile": (C.c_int, [C.c_void_p, C.POINTER(rt_profile)]),
    "rt_device_count": (C.c_int, []),
    "rt_last_error": (C.c_char_p, []),
    "rt_abi_version": (C.c_uint32, []),
    "rt_auto_pass_paths": (C.c_uint64, []),
}

_P = C.c_void_p
_D3 = C.POINTER(C.c_double)


def builder_symbols(prefix):
    """The scene-building entry points shared (by name and signature) by the host library
    (prefix 'rth_') and the test oracle (prefix 'orc_')."""
    return {
        prefix + "sphere": (_P, [_D3, C.c_double]),
        prefix + "sphere_both_roots": (_P, [_D3, C.c_double]),
        prefix + "sphere_timed": (_P, [C.c_double, C.c_double, _D3, C.c_double, _D3]),
        prefix + "rectangle": (_P, [_D3, _D3]),
        prefix + "new_box": (_P, [_D3, _D3]),
        prefix + "flipped": (_P, [_P]),
        prefix + "translated": (_P, [_P, _D3]),
        prefix + "rotated": (_P, [_P, C.c_int, C.c_double]),
        prefix + "constant_medium": (_P, [_P, C.c_double]),
        prefix + "constant_medium_multi": (_P, [_P, C.c_double]),
        prefix + "shape_free": (None, [_P]),
        prefix + "shape_bound": (None, [_P, _D3, _D3]),
        prefix + "tex_solid": (_P, [_D3]),
        prefix + "tex_checkered": (_P, [_P, _P]),
        prefix + "tex_noise": (_P, [C.c_double, C.c_void_p]),
        prefix + "tex_noise_fbm": (_P, [C.c_double, C.c_uint32, C.c_uint32, C.c_double, C.c_double]),
        prefix + "tex_noise_turbulence": (_P, [C.c_double, C.c_void_p, C.c_uint32, C.c_uint32, C.c_double, C.c_double, C.c_double]),
        prefix + "tex_image": (_P, [C.c_uint32, C.c_uint32, C.c_void_p]),
        prefix + "tex_free": (None, [_P]),
        prefix + "mat_lambertian": (_P, [_P]),
        prefix + "mat_metal": (_P, [_D3, C.c_double]),
        prefix + "mat_dielectric": (_P, [_D3, C.c_double]),
        prefix + "mat_diffuse_light": (_P, [_P]),
        prefix + "mat_isotropic": (_P, [_P]),
        prefix + "mat_face_normal": (_P, [_P]),
        prefix + "mat_free": (None, [_P]),
        prefix + "perm_table": (None, [C.c_uint32, C.c_void_p]),
        prefix + "scene_free": (None, [_P]),
    }


HOST_SYMBOLS = dict(builder_symbols("rth_"))
HOST_SYMBOLS.update({
    "rth_last_error": (C.c_char_p, []),
    "rth_tex_image_file": (_P, [C.c_char_p]),
    "rth_scene_new": (_P, [C.c_size_t, C.POINTER(_P), C.POINTER(_P), _D3, C.c_uint64, C.c_int]),
    "rth_scene_desc": (C.POINTER(rt_scene_desc), [_P]),
    "rth_example_scene": (_P, [C.c_char_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_int, C.POINTER(camera_options)]),
    "rth_camera_new": (C.c_int, [C.POINTER(camera_options), C.POINTER(rt_camera)]),
    "rth_pixels": (C.c_int, [_P, C.POINTER(camera_options), C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_void_p]),
    "rth_pixels_devices": (C.c_int, [_P, C.POINTER(camera_options), C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.POINTER(C.c_int),
                                     C.c_void_p]),
    "rth_write_image": (C.c_int, [C.c_char_p, C.c_uint32, C.c_uint32, C.c_void_p]),
})


def bind(lib, symbols):
    for name, (res, args) in symbols.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    return lib


_hip = None
_host = None


def hip_lib():
    """librt_hip.so with every rt_abi.h symbol bound.  Raises OSError if it has not been built."""
    global _hip
    if _hip is None:
        if not os.path.exists(HIP_LIB_PATH):
            raise OSError(f"{HIP_LIB_PATH} is missing: build it with `make -C raytracing_amd/csrc` (there is no CPU fallback)")
        _hip = bind(C.CDLL(HIP_LIB_PATH, mode=C.RTLD_GLOBAL), HIP_SYMBOLS)
    return _hip


def host_lib():
    global _host
    if _host is None:
        hip_lib()
        if not os.path.exists(HOST_LIB_PATH):
            raise OSError(f"{HOST_LIB_PATH} is missing: build it with `make -C raytracing_amd/csrc`")
        _host = bind(C.CDLL(HOST_LIB_PATH), HOST_SYMBOLS)
    return _host


class RtError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"rt_status {status}: {message}")
        self.status = status


def check(status):
    if status != RT_OK:
        raise RtError(status, hip_lib().rt_last_error().decode())
