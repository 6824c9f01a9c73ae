//! The public API of the `raytracing` crate with its render loop on an MI355X.
//!
//! Everything a scene script touches — `pixels`, `ImageOptions`, `Scene`, `Object`, `RayHit`, `camera::{Camera,
//! CameraOptions}`, `geometry::*`, `materials::Material`, `textures::Texture` — has the reference's name, fields and
//! signature, so the reference's `examples/*.rs` build against this crate unchanged.  What differs is behind them:
//! `Scene::new` flattens the objects (`flatten`), builds the reference's BVH on the host and uploads both through
//! `rt_scene_create`; `pixels` is one `rt_render` call (include/rt_abi.h, librt_hip.so).
//!
//! Environment: `RT_DEVICE` (HIP device index, default the current device), `RT_DEVICES` (render one frame on that many
//! devices, interleaved tiles, default 1), `RT_SEED` (64-bit render seed; default drawn from `rand::thread_rng()`, as
//! the reference's samples are).
pub mod geometry;
pub mod materials;
pub mod textures;

mod ffi;
mod flatten;

use camera::Camera;
use geometry::{Boundable, BoundingBox, Ray, Shape, Vec3};
use materials::Material;
use rand::Rng;
use std::sync::Arc;

fn env_number<T: std::str::FromStr>(name: &str) -> Option<T> {
    std::env::var(name).ok().and_then(|v| v.parse().ok())
}

/// Renders `scene` through `camera`: `3 * width * height` bytes, RGB, top row first (`src/lib.rs:18` of the reference).
pub fn pixels(camera: &Camera, scene: &Scene, image_options: ImageOptions) -> Vec<u8> {
    let ImageOptions { height, width, samples_per_pixel, max_reflections } = image_options;
    let io = ffi::RtImageOptions { height, width, samples_per_pixel: samples_per_pixel as u64, max_reflections: max_reflections as u64 };
    let cam = camera.abi();
    let seed: u64 = env_number("RT_SEED").unwrap_or_else(|| rand::thread_rng().gen());
    let n = 3 * width as usize * height as usize;
    let options = |rank: u32, world_size: u32| ffi::RtRenderOptions {
        seed,
        rank,
        world_size,
        tile_w: 0,
        tile_h: 0,
        samples_per_pass: 0,
        flags: 0,
        stream: std::ptr::null_mut(),
    };
    let n_devices: u32 = env_number("RT_DEVICES").unwrap_or(1).max(1);
    let mut rgb = vec![0u8; n];
    if n_devices == 1 {
        let st = unsafe { ffi::rt_render(scene.device.handle, &cam, &io, &options(0, 1), rgb.as_mut_ptr(), std::ptr::null_mut()) };
        assert!(st == 0, "rt_render: {}", ffi::last_error());
        return rgb;
    }
    // One host thread, one copy of the scene and one frame of means per device; every pixel is rendered by exactly one
    // device and is zero in the other frames, so the sum is exact and the image does not depend on the device count.
    let flat = &scene.flat;
    let frames: Vec<Vec<f64>> = std::thread::scope(|s| {
        let workers: Vec<_> = (0..n_devices)
            .map(|rank| {
                let (cam, io) = (&cam, &io);
                let opts = options(rank, n_devices);
                s.spawn(move || {
                    let device = DeviceScene::upload_on(flat, rank as i32);
                    let mut mean = vec![0f64; n];
                    let st = unsafe { ffi::rt_render(device.handle, cam, io, &opts, std::ptr::null_mut(), mean.as_mut_ptr()) };
                    assert!(st == 0, "rt_render on device {}: {}", rank, ffi::last_error());
                    mean
                })
            })
            .collect();
        workers.into_iter().map(|w| w.join().unwrap()).collect()
    });
    for (k, byte) in rgb.iter_mut().enumerate() {
        let v: f64 = frames.iter().map(|f| f[k]).sum();
        *byte = (v.max(0.0).min(1.0).sqrt() * 255.999) as u8;
    }
    rgb
}

/// Frame size and sampling effort of one `pixels` call (the reference's struct of the same name, `src/lib.rs:108-117`).
#[derive(Debug, Clone, Copy)]
pub struct ImageOptions {
    /// Rows of the frame.
    pub height: u32,
    /// Columns of the frame.
    pub width: u32,
    /// Camera paths traced, and averaged, per pixel.
    pub samples_per_pixel: usize,
    /// Longest path, in segments; a path still alive after that many contributes black.
    pub max_reflections: usize,
}

/// A scene handle of the device library; destroyed with the last clone of the `Scene` that owns it.
pub(crate) struct DeviceScene {
    pub(crate) handle: *mut ffi::RtScene,
}
// Rendering changes a handle (its workspace grows with the frame, it learns a schedule, it regroups its tree), so the
// library serialises the calls made on one handle: every entry point of include/rt_abi.h that takes a handle holds the
// handle's own lock for the duration of the call.  Calls from several threads on one `Scene` are therefore safe and run
// one after the other; `rt_scene_destroy` runs in `drop`, when no other reference exists.
unsafe impl Send for DeviceScene {}
unsafe impl Sync for DeviceScene {}

impl DeviceScene {
    pub(crate) fn upload(flat: &flatten::Flat) -> DeviceScene {
        DeviceScene::upload_on(flat, env_number("RT_DEVICE").unwrap_or(-1))
    }

    pub(crate) fn upload_on(flat: &flatten::Flat, device: i32) -> DeviceScene {
        let desc = flat.desc();
        let mut handle = std::ptr::null_mut();
        let st = unsafe { ffi::rt_scene_create(&desc, device, &mut handle) };
        assert!(st == 0, "rt_scene_create: {}", ffi::last_error());
        DeviceScene { handle }
    }
}

impl Drop for DeviceScene {
    fn drop(&mut self) {
        unsafe { ffi::rt_scene_destroy(self.handle) }
    }
}

#[derive(Clone)]
pub struct Scene {
    flat: Arc<flatten::Flat>,
    device: Arc<DeviceScene>,
    n_objects: usize,
    background_color: Vec3,
}

impl std::fmt::Debug for Scene {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "Scene {{ objects: {}, bvh nodes: {}, background_color: {:?} }}", self.n_objects, self.flat.bvh.len(), self.background_color)
    }
}

impl Scene {
    /// Creates a new scene.  The vector of objects is converted to a BVH tree — the reference's tree for the same `rng`
    /// stream: one `gen_range(0, 3)` per branch node, in pre-order.  Panics if `objects` is empty.
    pub fn new<R: Rng + ?Sized>(objects: Vec<Object>, background_color: Vec3, rng: &mut R) -> Self {
        let flat = flatten::Flat::from_objects(&objects, background_color, &mut || rng.gen_range(0, 3), false);
        let device = DeviceScene::upload(&flat);
        Scene { flat: Arc::new(flat), device: Arc::new(device), n_objects: objects.len(), background_color }
    }
}

/// One thing in a scene: where it is and what its surface does to a ray (`src/lib.rs:261-266` of the reference).
#[derive(Clone, Debug)]
pub struct Object {
    /// Geometry, wrappers included.
    pub shape: Shape,
    /// Surface behaviour.
    pub material: Material,
}

impl Boundable for Object {
    fn bound(&self) -> Option<BoundingBox> {
        self.shape.bound()
    }
}

/// What a material makes of a hit (`src/lib.rs:300-306` of the reference): the path goes on, or it ends in a colour.
pub enum RayHit {
    /// The path continues along `ray`; its throughput is multiplied by `attenuation`.
    Scattered { ray: Ray, attenuation: Vec3 },
    /// The path ends here with this radiance (lights; also a metal's grazing miss, with black).
    Emitted(Vec3),
}

pub mod camera {
    use super::geometry::*;

    /// Thin-lens camera with a shutter interval, in the derived form the render loop uses (`src/lib.rs:314-341`).
    #[derive(Debug, Clone, Copy)]
    pub struct Camera {
        aspect_ratio: f64,
        horizontal: Vec3,
        lens_radius: f64,
        lower_left_corner: Point3,
        onb: Onb,
        origin: Point3,
        shutter_close: f64,
        shutter_open: f64,
        vertical: Vec3,
        vfov: f64,
    }

    impl Camera {
        /// Viewport, lens and orthonormal frame from the user's description (`src/lib.rs:347-384`).
        pub fn new(options: CameraOptions) -> Self {
            let CameraOptions { aperture, aspect_ratio, focus_distance, looking_at, origin, shutter_close, shutter_open, vfov, vup } = options;
            let viewport_height = 2.0 * (vfov / 2.0).to_radians().tan();
            let viewport_width = viewport_height * aspect_ratio;
            let onb = Onb::from23(vup, origin - looking_at);
            let horizontal = *onb[0] * viewport_width * focus_distance;
            let vertical = *onb[1] * viewport_height * focus_distance;
            let lower_left_corner = origin - horizontal / 2.0 - vertical / 2.0 - *onb[2] * focus_distance;
            Camera { aspect_ratio, horizontal, lens_radius: aperture / 2.0, lower_left_corner, onb, origin, shutter_close, shutter_open, vertical, vfov }
        }

        /// The camera as the device library takes it.
        pub(crate) fn abi(&self) -> crate::ffi::RtCamera {
            let _ = (self.aspect_ratio, self.vfov);
            crate::ffi::RtCamera {
                origin: self.origin.0,
                lower_left_corner: self.lower_left_corner.0,
                horizontal: self.horizontal.0,
                vertical: self.vertical.0,
                u: (*self.onb[0]).0,
                v: (*self.onb[1]).0,
                lens_radius: self.lens_radius,
                shutter_open: self.shutter_open,
                shutter_close: self.shutter_close,
            }
        }
    }

    /// The user's description of a camera; same fields, order and meaning as the reference's (`src/lib.rs:405-425`).
    #[derive(Debug, Clone, Copy)]
    pub struct CameraOptions {
        /// Lens diameter (the lens radius is half of it); 0 is a pinhole.
        pub aperture: f64,
        /// Viewport width over height.
        pub aspect_ratio: f64,
        /// Distance from `origin` to the plane of sharp focus.
        pub focus_distance: f64,
        /// The view is centred on this point.
        pub looking_at: Point3,
        /// Position of the lens centre.
        pub origin: Point3,
        /// End of the exposure: ray times are drawn from `[shutter_open, shutter_close)`.
        pub shutter_close: f64,
        /// Start of the exposure.
        pub shutter_open: f64,
        /// Vertical opening angle of the view, degrees.
        pub vfov: f64,
        /// Which way is up in the picture (need not be orthogonal to the view direction).
        pub vup: Vec3,
    }
}
