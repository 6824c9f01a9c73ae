"""Python binding of the host API (include/raytracing.hpp) and of the C ABI (include/rt_abi.h).

The classes mirror the reference crate's public types (`Shape`, `Texture`, `Material`,
`Object`, `Scene`, `Camera`/`CameraOptions`, `ImageOptions`, `pixels`).  A Python
`Shape`/`Texture`/`Material` is a small recipe tree; it is instantiated through the C
builder entry points of a library (`rth_*` of the host library here, `orc_*` of the test
oracle in tests/), so one scene definition can feed both sides of a parity test.
All rendering goes through librt_hip.so; nothing here computes an image on the CPU.
"""
import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _abi

AXIS_X, AXIS_Y, AXIS_Z = 0, 1, 2


def _d3(v):
    return (C.c_double * 3)(float(v[0]), float(v[1]), float(v[2]))


class Backend:
    """Scene-building entry points of one library, addressed by prefix."""

    def __init__(self, lib, prefix):
        self.lib, self.prefix = lib, prefix

    def __getattr__(self, name):
        return getattr(self.lib, self.prefix + name)


def host_backend():
    return Backend(_abi.host_lib(), "rth_")


class Shape:
    """src/geometry/shape.rs:62-106 (recipe tree)."""

    def __init__(self, kind, **kw):
        self.kind, self.kw = kind, kw

    # constructors named as in the reference
    @staticmethod
    def Sphere(center, radius, both_roots=False):
        """`both_roots` is NOT the reference's behaviour: the sphere is also hit from within, at its far root (RT_PRIM_BOTH_ROOTS)."""
        return Shape("sphere", center=tuple(center), radius=float(radius), both_roots=bool(both_roots))

    @staticmethod
    def sphere(radius, start: Tuple[float, Sequence[float]], end: Tuple[float, Sequence[float]]):
        """Shape::sphere (shape.rs:222): raises ValueError if end_time <= start_time."""
        if not end[0] > start[0]:
            raise ValueError("Shape::sphere: end_time must be greater than start_time")
        return Shape("sphere_timed", radius=float(radius), t0=float(start[0]), c0=tuple(start[1]), t1=float(end[0]), c1=tuple(end[1]))

    @staticmethod
    def rectangle(lower_left, upper_right):
        return Shape("rectangle", ll=tuple(lower_left), ur=tuple(upper_right))

    @staticmethod
    def new_box(mn, mx):
        return Shape("new_box", mn=tuple(mn), mx=tuple(mx))

    @staticmethod
    def ConstantMedium(boundary, density, multi_scatter=False):
        """`multi_scatter` is NOT the reference's behaviour: the medium's boundary queries see the ray behind its origin where it is
        (RT_PRIM_MULTI_SCATTER), so a ray that starts inside the boundary scatters again; the reference's media scatter once per entry."""
        return Shape("constant_medium", inner=boundary, density=float(density), multi_scatter=bool(multi_scatter))

    def flipped(self):
        return Shape("flipped", inner=self)

    def translated(self, offset):
        return Shape("translated", inner=self, offset=tuple(offset))

    def rotated(self, axis, angle):
        return Shape("rotated", inner=self, axis=int(axis), angle=float(angle))

    rotate = rotated

    def build(self, be: Backend):
        """Instantiates the tree in a library; returns an owned handle (free with be.shape_free)."""
        k, kw = self.kind, self.kw
        if k == "sphere":
            return (be.sphere_both_roots if kw.get("both_roots") else be.sphere)(_d3(kw["center"]), kw["radius"])
        if k == "sphere_timed":
            h = be.sphere_timed(kw["radius"], kw["t0"], _d3(kw["c0"]), kw["t1"], _d3(kw["c1"]))
            if not h:
                raise ValueError("Shape::sphere rejected by the library")
            return h
        if k == "rectangle":
            return be.rectangle(_d3(kw["ll"]), _d3(kw["ur"]))
        if k == "new_box":
            return be.new_box(_d3(kw["mn"]), _d3(kw["mx"]))
        inner = kw["inner"].build(be)
        try:
            if k == "flipped":
                return be.flipped(inner)
            if k == "translated":
                return be.translated(inner, _d3(kw["offset"]))
            if k == "rotated":
                return be.rotated(inner, kw["axis"], kw["angle"])
            if k == "constant_medium":
                return (be.constant_medium_multi if kw.get("multi_scatter") else be.constant_medium)(inner, kw["density"])
        finally:
            be.shape_free(inner)
        raise ValueError(k)


class Texture:
    """src/textures.rs:8-19 (recipe tree)."""

    def __init__(self, kind, **kw):
        self.kind, self.kw = kind, kw

    @staticmethod
    def SolidColor(color):
        return Texture("solid", color=tuple(color))

    @staticmethod
    def Checkered(odd, even):
        return Texture("checkered", odd=odd, even=even)

    @staticmethod
    def Noise(scale, perm_table=None):
        """noise::Perlin with an explicit 256-entry permutation table (default: the seed-0 table)."""
        return Texture("noise", scale=float(scale), table=None if perm_table is None else bytes(perm_table))

    LACUNARITY = 2.0943951023931953  # 2 pi / 3, the noise crate's default

    @staticmethod
    def Fbm(scale, seed=0, octaves=6, lacunarity=LACUNARITY, persistence=0.5):
        """Noise { scale, noise: Fbm } with the noise crate's defaults; octave i is a Perlin of seed + i.  Not used by any example of
        the reference (an extension: textures.rs:14-17 takes any NoiseFn)."""
        return Texture("fbm", scale=float(scale), seed=int(seed), octaves=int(octaves), lacunarity=float(lacunarity), persistence=float(persistence))

    @staticmethod
    def Turbulence(scale, seed=0, roughness=3, power=1.0, source_table=None, lacunarity=LACUNARITY, persistence=0.5):
        """Noise { scale, noise: Turbulence<Perlin> }: the source Perlin (default: seed 0) at the point displaced by three Fbm
        functions of seeds seed, seed + 1, seed + 2 with `roughness` octaves.  An extension like Fbm."""
        return Texture("turbulence", scale=float(scale), seed=int(seed), roughness=int(roughness), power=float(power),
                       table=None if source_table is None else bytes(source_table), lacunarity=float(lacunarity), persistence=float(persistence))

    @staticmethod
    def Image(rgb: np.ndarray):
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        assert rgb.ndim == 3 and rgb.shape[2] == 3
        return Texture("image", rgb=rgb)

    def build(self, be: Backend):
        k, kw = self.kind, self.kw
        if k == "solid":
            return be.tex_solid(_d3(kw["color"]))
        if k == "checkered":
            o, e = kw["odd"].build(be), kw["even"].build(be)
            try:
                return be.tex_checkered(o, e)
            finally:
                be.tex_free(o)
                be.tex_free(e)
        if k == "noise":
            table = kw["table"]
            if table is None:
                buf = (C.c_uint8 * 256)()
                be.perm_table(0, buf)
            else:
                buf = (C.c_uint8 * 256).from_buffer_copy(table)
            return be.tex_noise(kw["scale"], buf)
        if k == "fbm":
            t = be.tex_noise_fbm(kw["scale"], kw["seed"], kw["octaves"], kw["lacunarity"], kw["persistence"])
            if not t:
                raise ValueError("Texture.Fbm: 1 to 32 octaves")
            return t
        if k == "turbulence":
            table = kw["table"]
            if table is None:
                buf = (C.c_uint8 * 256)()
                be.perm_table(0, buf)
            else:
                buf = (C.c_uint8 * 256).from_buffer_copy(table)
            t = be.tex_noise_turbulence(kw["scale"], buf, kw["seed"], kw["roughness"], kw["power"], kw["lacunarity"], kw["persistence"])
            if not t:
                raise ValueError("Texture.Turbulence: roughness 1 to 32")
            return t
        if k == "image":
            rgb = kw["rgb"]
            return be.tex_image(rgb.shape[1], rgb.shape[0], rgb.ctypes.data_as(C.c_void_p))
        raise ValueError(k)


class Material:
    """src/materials.rs:8-18."""

    def __init__(self, kind, texture=None, albedo=(0.0, 0.0, 0.0), param=0.0):
        self.kind, self.texture, self.albedo, self.param = kind, texture, tuple(albedo), float(param)

    @staticmethod
    def Lambertian(texture):
        return Material("lambertian", texture=texture)

    @staticmethod
    def Metal(albedo, fuzz):
        return Material("metal", albedo=albedo, param=fuzz)

    @staticmethod
    def Dielectric(albedo, refraction_index):
        return Material("dielectric", albedo=albedo, param=refraction_index)

    @staticmethod
    def DiffuseLight(emit):
        return Material("diffuse_light", texture=emit)

    @staticmethod
    def Isotropic(texture):
        return Material("isotropic", texture=texture)

    @staticmethod
    def glass():
        return Material.Dielectric((1.0, 1.0, 1.0), 1.5)

    def facing_normal(self):
        """NOT the reference's behaviour: this material, handed the normal turned against the incoming ray (RT_MATF_FACE_NORMAL)."""
        m = Material(self.kind, self.texture, self.albedo, self.param)
        m.face_normal = True
        return m

    def build(self, be: Backend):
        m = self._build(be)
        if getattr(self, "face_normal", False):
            plain, m = m, be.mat_face_normal(m)
            be.mat_free(plain)
        return m

    def _build(self, be: Backend):
        if self.kind in ("lambertian", "diffuse_light", "isotropic"):
            t = self.texture.build(be)
            try:
                return getattr(be, "mat_" + self.kind)(t)
            finally:
                be.tex_free(t)
        return getattr(be, "mat_" + self.kind)(_d3(self.albedo), self.param)


@dataclass
class Object:
    """src/lib.rs:261-266"""
    shape: Shape
    material: Material


@dataclass
class CameraOptions:
    """src/lib.rs:405-425"""
    aperture: float
    aspect_ratio: float
    focus_distance: float
    looking_at: Sequence[float]
    origin: Sequence[float]
    shutter_close: float
    shutter_open: float
    vfov: float
    vup: Sequence[float]

    def to_c(self):
        return _abi.camera_options(self.aperture, self.aspect_ratio, self.focus_distance, _d3(self.looking_at), _d3(self.origin), self.shutter_close,
                                   self.shutter_open, self.vfov, _d3(self.vup))

    @staticmethod
    def from_c(c):
        return CameraOptions(c.aperture, c.aspect_ratio, c.focus_distance, tuple(c.looking_at), tuple(c.origin), c.shutter_close, c.shutter_open, c.vfov,
                             tuple(c.vup))


class Camera:
    """Camera::new (src/lib.rs:347-384), computed by the host library."""

    def __init__(self, options: CameraOptions):
        self.options = options
        self.abi = _abi.rt_camera()
        co = options.to_c()
        if _abi.host_lib().rth_camera_new(C.byref(co), C.byref(self.abi)) != 0:
            raise ValueError(_abi.host_lib().rth_last_error().decode())


@dataclass
class ImageOptions:
    """src/lib.rs:108-117"""
    height: int
    width: int
    samples_per_pixel: int
    max_reflections: int

    def to_c(self):
        return _abi.rt_image_options(self.height, self.width, self.samples_per_pixel, self.max_reflections)


@dataclass
class RenderOptions:
    seed: int = 1
    rank: int = 0
    world_size: int = 1
    tile_w: int = 0
    tile_h: int = 0
    samples_per_pass: int = 0
    flags: int = 0
    stream: Optional[int] = None

    def to_c(self):
        # A stream handle of 0 is torch's default stream, but the ABI reads NULL as "the library's own stream", which has no
        # order with the default stream: refuse it rather than run unordered.  None asks for the library's stream by name.
        if self.stream is not None and int(self.stream) == 0:
            raise ValueError("stream handle 0 (the default stream) cannot be named through the ABI: pass None for the library's own stream, "
                             "or the handle of a torch.cuda.Stream")
        return _abi.rt_render_options(self.seed, self.rank, self.world_size, self.tile_w, self.tile_h, self.samples_per_pass, self.flags, self.stream)


def _current_device() -> int:
    """The device index torch has current in this thread (what tensors named "cuda" land on); 0 when torch is not in use."""
    import sys
    torch = sys.modules.get("torch")
    if torch is not None and torch.cuda.is_available():
        return int(torch.cuda.current_device())
    return 0


def rays_to_c(origins, directions, times):
    o = np.asarray(origins, dtype=np.float32).reshape(-1, 3)
    d = np.asarray(directions, dtype=np.float32).reshape(-1, 3)
    t = np.broadcast_to(np.asarray(times, dtype=np.float32), (o.shape[0],))
    arr = np.zeros((o.shape[0], 7), dtype=np.float32)
    arr[:, 0:3], arr[:, 3:6], arr[:, 6] = o, d, t
    return arr


HIT_DTYPE = np.dtype([("hit", np.int32), ("prim", np.uint32), ("t", np.float32), ("point", np.float32, 3), ("normal", np.float32, 3),
                      ("in_vec", np.float32, 3), ("u", np.float32), ("v", np.float32)])
assert HIT_DTYPE.itemsize == C.sizeof(_abi.rt_hit)
SCATTER_DTYPE = np.dtype([("scattered", np.int32), ("color", np.float32, 3), ("origin", np.float32, 3), ("direction", np.float32, 3)])
assert SCATTER_DTYPE.itemsize == C.sizeof(_abi.rt_scatter)


class Scene:
    """Scene::new (src/lib.rs:127): flattened by the host library; the device copy is made by rt_scene_create."""

    def __init__(self, objects: List[Object], background_color, seed: int = 1, keep_inert: bool = False, _handle=None):
        self._host = _abi.host_lib()
        self._hip = _abi.hip_lib()
        self._dev = {}
        if _handle is not None:
            self._h = _handle
            return
        be = host_backend()
        shapes = [o.shape.build(be) for o in objects]
        mats = [o.material.build(be) for o in objects]
        try:
            n = len(objects)
            self._h = self._host.rth_scene_new(n, (C.c_void_p * max(n, 1))(*shapes), (C.c_void_p * max(n, 1))(*mats), _d3(background_color), seed,
                                               int(keep_inert))
        finally:
            for s in shapes:
                be.shape_free(s)
            for m in mats:
                be.mat_free(m)
        if not self._h:
            raise ValueError(self._host.rth_last_error().decode())

    @staticmethod
    def example(name: str, scene_seed: int = 1, earthmap: Optional[np.ndarray] = None, keep_inert: bool = False):
        """One of the reference's five example scenes (examples/*.rs); returns (scene, CameraOptions)."""
        host = _abi.host_lib()
        co = _abi.camera_options()
        if earthmap is not None:
            earthmap = np.ascontiguousarray(earthmap, dtype=np.uint8)
            h = host.rth_example_scene(name.encode(), scene_seed, earthmap.shape[1], earthmap.shape[0], earthmap.ctypes.data_as(C.c_void_p), int(keep_inert),
                                       C.byref(co))
        else:
            h = host.rth_example_scene(name.encode(), scene_seed, 0, 0, None, int(keep_inert), C.byref(co))
        if not h:
            raise ValueError(host.rth_last_error().decode())
        return Scene(None, None, _handle=h), CameraOptions.from_c(co)

    @property
    def desc(self) -> _abi.rt_scene_desc:
        return self._host.rth_scene_desc(self._h).contents

    def device(self, device: int = -1):
        """rt_scene handle on HIP device `device` (-1: the device torch has current, else device 0), created on first use
        and kept per device.  Raises RtError without a GPU."""
        if device < 0:
            device = _current_device()
        if device not in self._dev:
            out = C.c_void_p()
            _abi.check(self._hip.rt_scene_create(self._host.rth_scene_desc(self._h), device, C.byref(out)))
            self._dev[device] = out
        return self._dev[device]

    def close(self):
        for dev in list(getattr(self, "_dev", None) or {}):
            self._hip.rt_scene_destroy(self._dev.pop(dev))
        if getattr(self, "_h", None):
            self._host.rth_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- C ABI calls
    def render(self, camera: Camera, image_options: ImageOptions, render_options: RenderOptions = None, want_mean: bool = False):
        """rt_render: returns rgb (h, w, 3) uint8 [, mean (h, w, 3) float64]."""
        ro = (render_options or RenderOptions()).to_c()
        io = image_options.to_c()
        rgb = np.zeros((image_options.height, image_options.width, 3), dtype=np.uint8)
        mean = np.zeros((image_options.height, image_options.width, 3), dtype=np.float64) if want_mean else None
        _abi.check(self._hip.rt_render(self.device(), C.byref(camera.abi), C.byref(io), C.byref(ro), rgb.ctypes.data_as(C.c_void_p),
                                       mean.ctypes.data_as(C.c_void_p) if want_mean else None))
        return (rgb, mean) if want_mean else rgb

    def render_device(self, camera: Camera, image_options: ImageOptions, render_options: RenderOptions, d_mean_ptr: int):
        """rt_render_device: enqueue the render of this rank's tiles into a device buffer of 3*w*h doubles."""
        ro = render_options.to_c()
        io = image_options.to_c()
        _abi.check(self._hip.rt_render_device(self.device(), C.byref(camera.abi), C.byref(io), C.byref(ro), C.c_void_p(d_mean_ptr)))

    def render_device_noalloc(self, camera: Camera, image_options: ImageOptions, render_options: RenderOptions, d_mean_ptr: int):
        """rt_render_device_noalloc: the same render out of the library's static device workspace (no allocation; size limits apply)."""
        ro = render_options.to_c()
        io = image_options.to_c()
        _abi.check(self._hip.rt_render_device_noalloc(self.device(), C.byref(camera.abi), C.byref(io), C.byref(ro), C.c_void_p(d_mean_ptr)))

    def quantise_device(self, d_mean_ptr: int, n: int, d_rgb_ptr: int, stream: Optional[int] = None):
        """rt_quantise_device on `stream` (a real stream handle; None = the library's own stream, see RenderOptions.to_c)."""
        if stream is not None and int(stream) == 0:
            raise ValueError("stream handle 0 (the default stream) cannot be named through the ABI")
        _abi.check(self._hip.rt_quantise_device(self.device(), C.c_void_p(d_mean_ptr), n, C.c_void_p(d_rgb_ptr), stream))

    def profile(self) -> dict:
        p = _abi.rt_profile()
        _abi.check(self._hip.rt_get_profile(self.device(), C.byref(p)))
        out = {"samples": p.samples, "segments": p.segments, "traverse_segments": p.traverse_segments, "passes": p.passes,
               "tree_tests_by_area": p.tree_tests_by_area, "tree_tests_sampled": p.tree_tests_sampled,
               "tail_from": p.tail_from, "two_stream_passes": p.two_stream_passes, "pass_paths": p.pass_paths,
               "two_bounce_passes": p.two_bounce_passes}
        for k, name in enumerate(_abi.RT_K_NAMES):
            out[name + "_ms"] = p.kernel_ms[k]
            out[name + "_launches"] = p.kernel_launches[k]
        return out

    def intersect_batch(self, origins, directions, times, tmin=0.001, tmax=float("inf"), scope=-1, seed=0, medium_u=None):
        """rt_intersect_batch: closest hit (scope=-1) or Intersection::intersect of one primitive; returns a HIT_DTYPE array."""
        rays = rays_to_c(origins, directions, times)
        n = rays.shape[0]
        out = np.zeros(n, dtype=HIT_DTYPE)
        mu = None if medium_u is None else np.ascontiguousarray(np.broadcast_to(np.asarray(medium_u, dtype=np.float32), (n,)))
        _abi.check(self._hip.rt_intersect_batch(self.device(), scope, rays.ctypes.data_as(C.POINTER(_abi.rt_ray)), n, tmin, tmax, seed,
                                                mu.ctypes.data_as(C.c_void_p) if mu is not None else None, out.ctypes.data_as(C.POINTER(_abi.rt_hit))))
        return out

    def scatter_batch(self, hits, seed=0, keys=None):
        """rt_scatter_batch: Material::scatter (materials.rs:28) on HIT_DTYPE records; keys = (n, 3) pixel / sample / bounce.
        Returns a SCATTER_DTYPE array."""
        hits = np.ascontiguousarray(hits, dtype=HIT_DTYPE)
        n = hits.shape[0]
        out = np.zeros(n, dtype=SCATTER_DTYPE)
        k = None if keys is None else np.ascontiguousarray(keys, dtype=np.uint32).reshape(n, 3)
        _abi.check(self._hip.rt_scatter_batch(self.device(), hits.ctypes.data_as(C.POINTER(_abi.rt_hit)), n, seed,
                                              k.ctypes.data_as(C.c_void_p) if k is not None else None, out.ctypes.data_as(C.POINTER(_abi.rt_scatter))))
        return out

    def camera_rays(self, camera: Camera, image_options: ImageOptions, seed, px_i, px_j, sample):
        px_i, px_j, sample = (np.ascontiguousarray(a, dtype=np.uint32) for a in (px_i, px_j, sample))
        n = px_i.shape[0]
        out = np.zeros((n, 7), dtype=np.float32)
        io = image_options.to_c()
        _abi.check(self._hip.rt_camera_rays(self.device(), C.byref(camera.abi), C.byref(io), seed, n, px_i.ctypes.data_as(C.c_void_p),
                                            px_j.ctypes.data_as(C.c_void_p), sample.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.POINTER(_abi.rt_ray))))
        return out


def pixels(camera: Camera, scene: Scene, image_options: ImageOptions, seed: int = 1, devices=None) -> bytes:
    """`pixels(&camera, &scene, image_options) -> Vec<u8>` (src/lib.rs:18) through the host library's C++ entry.
    `devices`: device indices to split the frame over (one host thread each); None = the current device."""
    host = _abi.host_lib()
    out = np.zeros(3 * image_options.width * image_options.height, dtype=np.uint8)
    co = camera.options.to_c()
    io = image_options
    if devices is None:
        rc = host.rth_pixels(scene._h, C.byref(co), io.width, io.height, io.samples_per_pixel, io.max_reflections, seed, -1, out.ctypes.data_as(C.c_void_p))
    else:
        dev = (C.c_int * len(devices))(*devices)
        rc = host.rth_pixels_devices(scene._h, C.byref(co), io.width, io.height, io.samples_per_pixel, io.max_reflections, seed, len(devices), dev,
                                     out.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise RuntimeError(host.rth_last_error().decode())
    return out.tobytes()
