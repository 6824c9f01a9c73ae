"""The Rust facade crate (rust/raytracing) cannot be compiled here: no Rust toolchain exists in this environment.
What can be checked without one is checked here: that its #[repr(C)] structs lay out exactly like the ctypes
structures the GPU tests drive the library with (names, order, offsets, sizes), that its constants carry the header's
values, that every extern "C" function it declares is exported by librt_hip.so with the same number of arguments, and
that the public items the reference's example programs import are all defined."""
import ctypes as C
import os
import re

import pytest

from raytracing_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CRATE = os.path.join(ROOT, "rust", "raytracing")
SRC = os.path.join(CRATE, "src")

PRIMITIVE = {"u8": (1, 1), "i8": (1, 1), "u16": (2, 2), "i16": (2, 2), "u32": (4, 4), "i32": (4, 4), "f32": (4, 4), "u64": (8, 8), "i64": (8, 8),
             "f64": (8, 8)}
STRUCT_OF = {"RtPrim": _abi.rt_prim, "RtSphere": _abi.rt_sphere, "RtRect": _abi.rt_rect, "RtBox": _abi.rt_box, "RtXform": _abi.rt_xform,
             "RtMaterial": _abi.rt_material, "RtTexture": _abi.rt_texture, "RtImage": _abi.rt_image, "RtBvhNode": _abi.rt_bvh_node,
             "RtSceneDesc": _abi.rt_scene_desc, "RtCamera": _abi.rt_camera, "RtImageOptions": _abi.rt_image_options,
             "RtRenderOptions": _abi.rt_render_options, "RtRay": _abi.rt_ray, "RtHit": _abi.rt_hit, "RtScatter": _abi.rt_scatter}


def read(*parts):
    with open(os.path.join(*parts)) as f:
        return f.read()


def strip_comments(text):
    return re.sub(r"//[^\n]*", "", text)


def size_align(ty):
    ty = ty.strip()
    if ty.startswith("*"):
        return 8, 8
    m = re.fullmatch(r"\[\s*(\w+)\s*;\s*(\d+)\s*\]", ty)
    if m:
        s, a = PRIMITIVE[m.group(1)]
        return s * int(m.group(2)), a
    return PRIMITIVE[ty]


def rust_structs():
    text = strip_comments(read(SRC, "ffi.rs"))
    out = {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*(?:#\[derive\([^)]*\)\]\s*)*pub struct (\w+)\s*\{([^}]*)\}", text):
        fields, offset, max_align = [], 0, 1
        for f in re.finditer(r"pub (\w+)\s*:\s*([^,]+),", m.group(2) + ","):
            size, align = size_align(f.group(2))
            offset = (offset + align - 1) // align * align
            fields.append((f.group(1), offset, size))
            offset += size
            max_align = max(max_align, align)
        out[m.group(1)] = (fields, (offset + max_align - 1) // max_align * max_align)
    return out


def test_repr_c_structs_match_the_ctypes_layouts():
    structs = rust_structs()
    assert set(STRUCT_OF) <= set(structs), sorted(set(STRUCT_OF) - set(structs))
    for name, ctype in STRUCT_OF.items():
        fields, size = structs[name]
        expect = [(fname, getattr(ctype, fname).offset, getattr(ctype, fname).size) for fname, _ in ctype._fields_]
        assert fields == expect, name
        assert size == C.sizeof(ctype), name


def test_constants_match_the_header():
    header = read(ROOT, "include", "rt_abi.h")
    values = {}
    for m in re.finditer(r"\b(RT_[A-Z0-9_]+)\s*=\s*(0x[0-9A-Fa-f]+|\d+)u?\b", header):
        values[m.group(1)] = int(m.group(2), 0)
    for m in re.finditer(r"#define\s+(RT_[A-Z0-9_]+)\s+(0x[0-9A-Fa-f]+|\d+)u?\b", header):
        values[m.group(1)] = int(m.group(2), 0)
    rust = {m.group(1): int(m.group(2).replace("_", ""), 0)
            for m in re.finditer(r"pub const (RT_[A-Z0-9_]+)\s*:\s*\w+\s*=\s*(0x[0-9A-Fa-f_]+|\d+)\s*;", read(SRC, "ffi.rs"))}
    assert len(rust) >= 20
    for name, value in rust.items():
        assert name in values, name
        assert values[name] == value, name


def test_extern_functions_are_exported_with_the_same_arity():
    text = strip_comments(read(SRC, "ffi.rs"))
    block = re.search(r'extern "C" \{(.*?)\n\}', text, re.S).group(1)
    found = {m.group(1): m.group(2) for m in re.finditer(r"pub fn (\w+)\s*\(([^)]*)\)", block)}
    assert {"rt_scene_create", "rt_scene_destroy", "rt_render", "rt_intersect_batch", "rt_scatter_batch", "rt_last_error"} <= set(found)
    lib = C.CDLL(_abi.HIP_LIB_PATH)
    for name, args in found.items():
        assert name in _abi.HIP_SYMBOLS, name
        getattr(lib, name)  # exported
        n_args = len([a for a in args.split(",") if a.strip()])
        assert n_args == len(_abi.HIP_SYMBOLS[name][1]), name


# What the reference's five example programs import and name (their `use raytracing::...` lines and the variants and
# fields they construct).
PUBLIC_ITEMS = {
    "lib.rs": [r"pub fn pixels\(camera: &Camera, scene: &Scene, image_options: ImageOptions\) -> Vec<u8>", r"pub struct ImageOptions",
               r"pub samples_per_pixel: usize", r"pub max_reflections: usize", r"pub struct Scene",
               r"pub fn new<R: Rng \+ \?Sized>\(objects: Vec<Object>, background_color: Vec3, rng: &mut R\) -> Self", r"pub struct Object",
               r"pub shape: Shape", r"pub material: Material", r"pub enum RayHit", r"pub mod camera", r"pub struct Camera\b",
               r"pub struct CameraOptions", r"pub aperture: f64", r"pub aspect_ratio: f64", r"pub focus_distance: f64", r"pub looking_at: Point3",
               r"pub origin: Point3", r"pub shutter_close: f64", r"pub shutter_open: f64", r"pub vfov: f64", r"pub vup: Vec3",
               r"pub mod geometry;", r"pub mod materials;", r"pub mod textures;"],
    "geometry.rs": [r"pub enum Axis", r"triple!\(Point3\)", r"triple!\(Vec3\)", r"pub struct UnitVec3", r"pub struct Onb", r"pub struct Ray",
                    r"pub fn random<R: Rng \+ \?Sized>\(range: Range<f64>, rng: &mut R\)", r"pub fn dist\(&self", r"pub fn normed\(&self\)",
                    r"pub struct InsideUnitSphere", r"pub struct OnUnitSphere", r"pub struct InsideUnitDisc", r"pub fn random_unit_vector"],
    os.path.join("geometry", "shape.rs"): [r"pub enum Shape", r"Sphere \{ center: Point3, radius: f64 \}", r"pub fn rectangle\(lower_left: Point3, upper_right: Point3\)",
                                           r"pub fn new_box\(min: Point3, max: Point3\)", r"pub fn sphere\(radius: f64", r"pub fn flipped\(self\)",
                                           r"pub fn translated\(self, offset: Vec3\)", r"pub fn rotated\(self, axis: Axis, angle: f64\)",
                                           r"ConstantMedium \{ boundary: Box<Shape>, density: f64 \}", r"pub trait Intersection", r"pub trait Boundable",
                                           r"pub struct IntersectionPoint", r"pub struct BoundingBox", r"pub enum Face"],
    "materials.rs": [r"pub enum Material", r"Lambertian \{ texture: Texture \}", r"Metal \{ albedo: Vec3, fuzz: f64 \}",
                     r"Dielectric \{ albedo: Vec3, refraction_index: f64 \}", r"DiffuseLight \{ emit: Texture \}", r"Isotropic \{ texture: Texture \}",
                     r"pub fn glass\(\) -> Self", r"pub fn scatter<R: Rng \+ \?Sized>\(&self, p: &IntersectionPoint, rng: &mut R\) -> RayHit"],
    "textures.rs": [r"pub enum Texture", r"SolidColor\(Vec3\)", r"Checkered \{ odd: Box<Texture>, even: Box<Texture> \}",
                    r"Noise \{ scale: f64, noise: Arc<dyn NoiseFn<\[f64; 3\]> \+ Sync \+ Send> \}", r"Image\(RgbImage\)",
                    r"pub fn image<P: AsRef<std::path::Path>>\(path: P\) -> Self", r"pub fn color_at\(&self, u: f64, v: f64, p: &Point3\) -> Vec3"],
}


@pytest.mark.parametrize("path", sorted(PUBLIC_ITEMS))
def test_public_api_items_are_defined(path):
    text = read(SRC, path)
    for pattern in PUBLIC_ITEMS[path]:
        assert re.search(pattern, text), pattern


@pytest.mark.parametrize("path", ["lib.rs", "ffi.rs", "flatten.rs", "geometry.rs", os.path.join("geometry", "shape.rs"), "materials.rs", "textures.rs",
                                  os.path.join("..", "build.rs")])
def test_sources_are_balanced(path):
    """A cheap guard against truncated or mangled files: brackets pair up outside strings, chars and comments."""
    text = strip_comments(read(SRC, path))
    text = re.sub(r'"(?:\\[\s\S]|[^"\\])*"', '""', text)
    text = re.sub(r"'(?:\\.|[^'\\])'", "''", text)
    stack, pairs = [], {")": "(", "]": "[", "}": "{"}
    for ch in text:
        if ch in "([{":
            stack.append(ch)
        elif ch in pairs:
            assert stack and stack.pop() == pairs[ch], path
    assert not stack, path


def test_facade_restates_the_same_perlin_table_and_tree_rules():
    """The constants the facade shares with the C++ host layer, which the GPU tests do exercise."""
    tex = read(SRC, "textures.rs")
    host = read(ROOT, "raytracing_amd", "csrc", "host", "raytracing.cpp")
    assert "3.889_855_325_553_107_4" in tex and "3.8898553255531074" in read(ROOT, "raytracing_amd", "csrc", "rt_core.h")
    flat = read(SRC, "flatten.rs")
    assert "4e-6 * max_abs" in flat and "4e-6 * max_abs" in host
    assert "s = [1u32, seed, seed, seed]" in tex and "s[4] = {1u, seed, seed, seed}" in host


def test_facade_carries_the_reference_s_generic_impls_and_the_noise_kinds():
    """Public impls of the reference that scene code can lean on (geometry.rs:52-64, :131-143; shape.rs:17-34), and the noise
    functions beyond Perlin mapped with the constants and shifts the kernels use."""
    geo, shape, tex, flat = read(SRC, "geometry.rs"), read(SRC, "geometry", "shape.rs"), read(SRC, "textures.rs"), read(SRC, "flatten.rs")
    assert re.search(r"impl Deref for \$name \{\s*type Target = \[f64\];", geo)
    assert re.search(r"impl<I: Intersection> Intersection for Vec<I>", shape) and re.search(r"impl<T: Boundable> Boundable for Vec<T>", shape)
    core = read(ROOT, "raytracing_amd", "csrc", "rt_core.h")
    for k in ("12414.0", "65124.0", "31337.0", "26519.0", "18128.0", "60493.0", "53820.0", "11213.0", "44845.0"):
        assert k in tex and k + "f" in core, k
    assert "RT_NOISE_FBM | (octaves << 8)" in flat and "RT_NOISE_TURBULENCE | (roughness << 8)" in flat
    ffi = read(SRC, "ffi.rs")
    assert re.search(r"RT_ABI_VERSION: u32 = 3;", ffi) and re.search(r"RT_PRIM_MULTI_SCATTER: u8 = 16;", ffi)
    # the facade's claim that a scene handle may be shared between threads rests on the library's per-handle lock
    assert "std::lock_guard<std::mutex> lock(sc->mutex)" in read(ROOT, "raytracing_amd", "csrc", "rt_hip.hip")


def test_cargo_manifest_names_the_crate_and_its_dependencies():
    manifest = read(CRATE, "Cargo.toml")
    assert re.search(r'^name = "raytracing"$', manifest, re.M)
    for dep in ("image", "rand", "noise"):
        assert re.search(rf'^{dep} = "', manifest, re.M), dep
