//! `#[repr(C)]` mirrors of include/rt_abi.h, field for field (tests/test_rust_facade.py checks names, order and widths
//! against the ctypes declarations the GPU tests use), and the entry points of librt_hip.so.
#![allow(dead_code)]
use std::os::raw::{c_char, c_void};

pub const RT_ABI_VERSION: u32 = 3;

pub const RT_GEOM_SPHERE: u8 = 0;
pub const RT_GEOM_MOVING_SPHERE: u8 = 1;
pub const RT_GEOM_RECT: u8 = 2;
pub const RT_GEOM_BOX: u8 = 3;

pub const RT_PRIM_FLIP: u8 = 1;
pub const RT_PRIM_MEDIUM: u8 = 2;
pub const RT_PRIM_INERT: u8 = 4;
pub const RT_PRIM_BOTH_ROOTS: u8 = 8; // not the reference's behaviour; the facade never sets it
pub const RT_PRIM_MULTI_SCATTER: u8 = 16; // not the reference's behaviour either (media that scatter more than once per entry); never set here
pub const RT_MATF_FACE_NORMAL: u32 = 1; // likewise (rt_material.reserved[1])

pub const RT_NOISE_PERLIN: u32 = 0;
pub const RT_NOISE_FBM: u32 = 1;
pub const RT_NOISE_TURBULENCE: u32 = 2;

pub const RT_XF_TRANSLATE: u32 = 0;
pub const RT_XF_ROTATE_X: u32 = 1;

pub const RT_MAT_LAMBERTIAN: u32 = 0;
pub const RT_MAT_METAL: u32 = 1;
pub const RT_MAT_DIELECTRIC: u32 = 2;
pub const RT_MAT_DIFFUSE_LIGHT: u32 = 3;
pub const RT_MAT_ISOTROPIC: u32 = 4;

pub const RT_TEX_SOLID: u32 = 0;
pub const RT_TEX_CHECKERED: u32 = 1;
pub const RT_TEX_NOISE: u32 = 2;
pub const RT_TEX_IMAGE: u32 = 3;

pub const RT_BVH_NONE: u32 = 0xFFFF_FFFF;
pub const RT_BVH_EXACT: u32 = 0x8000_0000;
pub const RT_MAX_XFORMS_PER_PRIM: usize = 8;

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct RtPrim {
    pub geom: u8,
    pub flags: u8,
    pub xf_count: u16,
    pub xf_first: u32,
    pub geom_index: u32,
    pub material: u32,
    pub density: f32,
    pub xf_outer: u32,
    pub reserved: [u32; 2],
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct RtSphere {
    pub center: [f64; 3],
    pub radius: f64,
    pub motion: [f64; 3],
    pub start_time: f64,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct RtRect {
    pub axis: u32,
    pub reserved: u32,
    pub x0: f64,
    pub y0: f64,
    pub x1: f64,
    pub y1: f64,
    pub height: f64,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct RtBox {
    pub min: [f64; 3],
    pub max: [f64; 3],
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct RtXform {
    pub kind: u32,
    pub reserved: u32,
    pub a: f64,
    pub b: f64,
    pub c: f64,
}

#[repr(C)]
#[derive(Clone, Copy, Default, PartialEq)]
pub struct RtMaterial {
    pub kind: u32,
    pub texture: u32,
    pub albedo: [f32; 3],
    pub param: f32,
    pub reserved: [u32; 2],
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct RtTexture {
    pub kind: u32,
    pub color: [f32; 3],
    pub odd: u32,
    pub even: u32,
    pub scale: f32,
    pub resource: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct RtImage {
    pub width: u32,
    pub height: u32,
    pub texel_offset: u64,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct RtBvhNode {
    pub min: [f32; 3],
    pub skip: u32,
    pub max: [f32; 3],
    pub prim: u32,
}

#[repr(C)]
pub struct RtSceneDesc {
    pub abi_version: u32,
    pub n_prims: u32,
    pub n_spheres: u32,
    pub n_rects: u32,
    pub n_boxes: u32,
    pub n_xforms: u32,
    pub n_materials: u32,
    pub n_textures: u32,
    pub n_images: u32,
    pub n_perm_tables: u32,
    pub n_bvh_nodes: u32,
    pub reserved0: u32,
    pub prims: *const RtPrim,
    pub spheres: *const RtSphere,
    pub rects: *const RtRect,
    pub boxes: *const RtBox,
    pub xforms: *const RtXform,
    pub materials: *const RtMaterial,
    pub textures: *const RtTexture,
    pub images: *const RtImage,
    pub texels: *const u8,
    pub texel_bytes: u64,
    pub perm_tables: *const u8,
    pub bvh: *const RtBvhNode,
    pub background: [f64; 3],
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct RtCamera {
    pub origin: [f64; 3],
    pub lower_left_corner: [f64; 3],
    pub horizontal: [f64; 3],
    pub vertical: [f64; 3],
    pub u: [f64; 3],
    pub v: [f64; 3],
    pub lens_radius: f64,
    pub shutter_open: f64,
    pub shutter_close: f64,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtImageOptions {
    pub height: u32,
    pub width: u32,
    pub samples_per_pixel: u64,
    pub max_reflections: u64,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtRenderOptions {
    pub seed: u64,
    pub rank: u32,
    pub world_size: u32,
    pub tile_w: u32,
    pub tile_h: u32,
    pub samples_per_pass: u32,
    pub flags: u32,
    pub stream: *mut c_void,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct RtRay {
    pub origin: [f32; 3],
    pub direction: [f32; 3],
    pub time: f32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct RtHit {
    pub hit: i32,
    pub prim: u32,
    pub t: f32,
    pub point: [f32; 3],
    pub normal: [f32; 3],
    pub in_vec: [f32; 3],
    pub u: f32,
    pub v: f32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct RtScatter {
    pub scattered: i32,
    pub color: [f32; 3],
    pub origin: [f32; 3],
    pub direction: [f32; 3],
}

/// Opaque scene handle of the library.
pub enum RtScene {}

extern "C" {
    pub fn rt_scene_create(desc: *const RtSceneDesc, device: i32, out: *mut *mut RtScene) -> i32;
    pub fn rt_scene_destroy(scene: *mut RtScene);
    pub fn rt_render(scene: *mut RtScene, camera: *const RtCamera, image_options: *const RtImageOptions, render_options: *const RtRenderOptions,
                     rgb_out: *mut u8, mean_out: *mut f64) -> i32;
    pub fn rt_render_device(scene: *mut RtScene, camera: *const RtCamera, image_options: *const RtImageOptions, render_options: *const RtRenderOptions,
                            d_mean_out: *mut f64) -> i32;
    pub fn rt_render_device_noalloc(scene: *mut RtScene, camera: *const RtCamera, image_options: *const RtImageOptions,
                                    render_options: *const RtRenderOptions, d_mean_out: *mut f64) -> i32;
    pub fn rt_quantise_device(scene: *mut RtScene, d_mean: *const f64, n: u64, d_rgb_out: *mut u8, stream: *mut c_void) -> i32;
    pub fn rt_intersect_batch(scene: *mut RtScene, scope: i64, rays: *const RtRay, n: u64, tmin: f32, tmax: f32, seed: u64, medium_u: *const f32,
                              out: *mut RtHit) -> i32;
    pub fn rt_scatter_batch(scene: *mut RtScene, hits: *const RtHit, n: u64, seed: u64, keys: *const u32, out: *mut RtScatter) -> i32;
    pub fn rt_device_count() -> i32;
    pub fn rt_last_error() -> *const c_char;
    pub fn rt_abi_version() -> u32;
    pub fn rt_auto_pass_paths() -> u64;
}

/// The library's message for the last failed call on this thread.
pub fn last_error() -> String {
    unsafe {
        let p = rt_last_error();
        if p.is_null() {
            String::new()
        } else {
            std::ffi::CStr::from_ptr(p).to_string_lossy().into_owned()
        }
    }
}
