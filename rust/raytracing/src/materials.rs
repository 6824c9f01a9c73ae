//! The materials of the public API (the reference's `src/materials.rs`): same enum, same variants and fields.
//! `scatter` is answered by the device code the renderer shades with, through the parity hook `rt_scatter_batch`.
use super::geometry::{IntersectionPoint, Point3, Ray, Shape, Vec3};
use crate::ffi;
use crate::flatten::Flat;
use crate::textures::Texture;
use crate::RayHit;
use rand::Rng;

#[derive(Clone, Debug)]
pub enum Material {
    Lambertian { texture: Texture },

    Metal { albedo: Vec3, fuzz: f64 },

    Dielectric { albedo: Vec3, refraction_index: f64 },

    DiffuseLight { emit: Texture },

    Isotropic { texture: Texture },
}

impl Material {
    pub fn glass() -> Self {
        Material::Dielectric { refraction_index: 1.5, albedo: Vec3([1.0, 1.0, 1.0]) }
    }

    /// One shading event at the intersection `p`.  The device draws from its counter-based generator; `rng` supplies the
    /// 64-bit key of that stream, so two calls with the same generator state give the same result.
    pub fn scatter<R: Rng + ?Sized>(&self, p: &IntersectionPoint, rng: &mut R) -> RayHit {
        // a one-object scene that carries this material; the object's own shape plays no part in shading
        let carrier = crate::Object { shape: Shape::Sphere { center: Point3::default(), radius: 1.0 }, material: self.clone() };
        let flat = Flat::from_objects(std::slice::from_ref(&carrier), Vec3::default(), &mut || 0usize, true);
        let scene = crate::DeviceScene::upload(&flat);
        let f = |v: [f64; 3]| [v[0] as f32, v[1] as f32, v[2] as f32];
        let hit = ffi::RtHit {
            hit: 1,
            prim: 0,
            t: p.t as f32,
            point: f(p.point.0),
            normal: f((*p.normal).0),
            in_vec: f((*p.in_vec).0),
            u: p.surface_coordinates.0 as f32,
            v: p.surface_coordinates.1 as f32,
        };
        let mut out = ffi::RtScatter::default();
        let seed: u64 = rng.gen();
        let st = unsafe { ffi::rt_scatter_batch(scene.handle, &hit, 1, seed, std::ptr::null(), &mut out) };
        assert!(st == 0, "rt_scatter_batch: {}", ffi::last_error());
        let d = |v: [f32; 3]| [v[0] as f64, v[1] as f64, v[2] as f64];
        if out.scattered != 0 {
            RayHit::Scattered { ray: Ray { origin: Point3(d(out.origin)), direction: Vec3(d(out.direction)) }, attenuation: Vec3(d(out.color)) }
        } else {
            RayHit::Emitted(Vec3(d(out.color)))
        }
    }
}
