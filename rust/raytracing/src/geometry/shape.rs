//! The shapes of the public API (the reference's `src/geometry/shape.rs`): same enum, same variants and fields, same
//! constructors.  `bound()` is computed here because the BVH is built on the host (`crate::flatten`); `intersect()` is
//! answered by the device library through its parity hook `rt_intersect_batch`.
use super::{Axis, Point3, Ray, Rotate, UnitVec3, Vec3};
use crate::ffi;
use crate::flatten::Flat;
use std::ops::Add;

/// Things that can be intersected by rays.
pub trait Intersection {
    /// The intersection of `ray` with `self` whose parameter lies in `interval`, if there is one; `time` is the time at
    /// which the ray is emitted.
    fn intersect(&self, ray: &Ray, interval: (f64, f64), time: f64) -> Option<IntersectionPoint>;
}

/// A list of things is hit where its nearest member is hit (the reference's impl, shape.rs:17-34, which `Shape::Box` uses for
/// its six sides): the members are asked in order, each within what is left of the interval after the hits before it, so of
/// two members hit at the same parameter the later one is reported (a rectangle accepts the end of its interval).
impl<I: Intersection> Intersection for Vec<I> {
    fn intersect(&self, ray: &Ray, interval: (f64, f64), time: f64) -> Option<IntersectionPoint> {
        let (lower, mut upper) = interval;
        let mut nearest = None;
        for member in self {
            if let Some(found) = member.intersect(ray, (lower, upper), time) {
                upper = found.t;
                nearest = Some(found);
            }
        }
        nearest
    }
}

/// Things that may have a bounding box.
pub trait Boundable {
    fn bound(&self) -> Option<BoundingBox>;
}

impl<T: Boundable> Boundable for Vec<T> {
    fn bound(&self) -> Option<BoundingBox> {
        let mut all: Option<BoundingBox> = None;
        for item in self {
            let b = item.bound()?;
            all = Some(match all {
                Some(a) => a + b,
                None => b,
            });
        }
        all
    }
}

#[derive(Clone, Debug)]
pub enum Shape {
    Sphere { center: Point3, radius: f64 },
    MovingSphere { ray: Ray, radius: f64, start_time: f64, end_time: f64 },
    Rectangle { axis: Axis, lower_left: (f64, f64), upper_right: (f64, f64), height: f64 },
    ConstantMedium { boundary: Box<Shape>, density: f64 },
    Box { min: Point3, max: Point3, sides: Vec<Shape> },
    Flipped(Box<Shape>),
    Translated { inner: Box<Shape>, offset: Vec3 },
    Rotated { axis: Axis, inner: Box<Shape>, angle: f64 },
}

impl Shape {
    pub fn flipped(self) -> Self {
        Shape::Flipped(Box::new(self))
    }

    pub fn translated(self, offset: Vec3) -> Self {
        Shape::Translated { inner: Box::new(self), offset }
    }

    /// `self` rotated by `angle` degrees about `axis`.
    pub fn rotated(self, axis: Axis, angle: f64) -> Self {
        self.rotate(axis, angle)
    }

    /// The axis-aligned rectangle spanned by two corners that share one coordinate (x is looked at first, then y).
    pub fn rectangle(lower_left: Point3, upper_right: Point3) -> Self {
        let axis = if lower_left[0] == upper_right[0] {
            Axis::X
        } else if lower_left[1] == upper_right[1] {
            Axis::Y
        } else {
            Axis::Z
        };
        let (p1, p2, o) = axis.rect_frame();
        Shape::Rectangle {
            axis,
            lower_left: (lower_left[p1], lower_left[p2]),
            upper_right: (upper_right[p1], upper_right[p2]),
            height: lower_left[o],
        }
    }

    /// The axis-aligned box between two corners: six rectangles in the order x0, x1, y0, y1, z0, z1, the lower one of each
    /// pair flipped.  The y sides are given the (x, z) extents in that order, exactly as the reference gives them.
    pub fn new_box(min: Point3, max: Point3) -> Self {
        let lo = min.0;
        let hi = max.0;
        // (axis, the two coordinates whose extents the side carries, in the order they are stored)
        let layout = [(Axis::X, 1, 2), (Axis::Y, 0, 2), (Axis::Z, 0, 1)];
        let mut sides = Vec::with_capacity(6);
        for &(axis, a, b) in layout.iter() {
            let o = axis.rect_frame().2;
            for &(height, flip) in [(lo[o], true), (hi[o], false)].iter() {
                let side = Shape::Rectangle { axis, lower_left: (lo[a], lo[b]), upper_right: (hi[a], hi[b]), height };
                sides.push(if flip { side.flipped() } else { side });
            }
        }
        Shape::Box { min, max, sides }
    }

    /// A sphere that moves from `start_center` to `end_center` between the two times; a plain `Sphere` when the two
    /// centres are equal.  Panics if `end_time <= start_time`.
    pub fn sphere(radius: f64, (start_time, start_center): (f64, Point3), (end_time, end_center): (f64, Point3)) -> Self {
        assert!(end_time > start_time);
        if start_center == end_center {
            Shape::Sphere { center: start_center, radius }
        } else {
            Shape::MovingSphere { ray: Ray { origin: start_center, direction: end_center - start_center }, radius, start_time, end_time }
        }
    }
}

impl Rotate for Shape {
    fn rotate(self, axis: Axis, angle: f64) -> Self {
        Shape::Rotated { axis, inner: Box::new(self), angle }
    }
}

fn cube(center: Point3, r: f64) -> BoundingBox {
    let e = Vec3([1.0, 1.0, 1.0]) * r;
    BoundingBox { min: center - e, max: center + e }
}

impl Boundable for Shape {
    fn bound(&self) -> Option<BoundingBox> {
        match self {
            Shape::Sphere { center, radius } => Some(cube(*center, *radius)),
            Shape::MovingSphere { ray, radius, start_time, end_time } => Some(cube(ray.at(0.0), *radius) + cube(ray.at(end_time - start_time), *radius)),
            Shape::Rectangle { axis, lower_left, upper_right, height } => {
                let (p1, p2, o) = axis.rect_frame();
                let (mut min, mut max) = (Point3::default(), Point3::default());
                min[p1] = lower_left.0;
                min[p2] = lower_left.1;
                min[o] = height - 0.0001;
                max[p1] = upper_right.0;
                max[p2] = upper_right.1;
                max[o] = height + 0.0001;
                Some(BoundingBox { min, max })
            }
            Shape::Box { min, max, .. } => Some(BoundingBox { min: *min, max: *max }),
            Shape::Flipped(inner) => inner.bound(),
            Shape::ConstantMedium { boundary, .. } => boundary.bound(),
            Shape::Translated { inner, offset } => inner.bound().map(|b| BoundingBox { min: b.min + *offset, max: b.max + *offset }),
            Shape::Rotated { axis, inner, angle } => inner.bound().map(|b| {
                let mut min = Point3([f64::INFINITY; 3]);
                let mut max = Point3([f64::NEG_INFINITY; 3]);
                for corner in 0..8u32 {
                    // the corner as the reference forms it: i * max + (1 - i) * min per coordinate
                    let mut p = [0.0; 3];
                    for c in 0..3 {
                        let i = ((corner >> (2 - c)) & 1) as f64;
                        p[c] = i * b.max[c] + (1.0 - i) * b.min[c];
                    }
                    let q = Vec3(p).rotate(*axis, *angle);
                    for c in 0..3 {
                        min[c] = min[c].min(q[c]);
                        max[c] = max[c].max(q[c]);
                    }
                }
                BoundingBox { min, max }
            }),
        }
    }
}

impl Intersection for Shape {
    /// Answered by the device: a one-object scene is uploaded and `rt_intersect_batch` queried with `scope = 0`.  The ray
    /// crosses the boundary in fp32, so the record agrees with the reference's f64 arithmetic to about 1e-7 relative.  A
    /// `ConstantMedium` draws its free-flight variate from the library's counter-based generator (seed 0).
    fn intersect(&self, ray: &Ray, interval: (f64, f64), time: f64) -> Option<IntersectionPoint> {
        let object = crate::Object { shape: self.clone(), material: crate::materials::Material::glass() };
        let flat = Flat::from_objects(std::slice::from_ref(&object), Vec3::default(), &mut || 0usize, true);
        let scene = crate::DeviceScene::upload(&flat);
        let f = |v: [f64; 3]| [v[0] as f32, v[1] as f32, v[2] as f32];
        let r = ffi::RtRay { origin: f(ray.origin.0), direction: f(ray.direction.0), time: time as f32 };
        let mut hit = ffi::RtHit::default();
        let st = unsafe { ffi::rt_intersect_batch(scene.handle, 0, &r, 1, interval.0 as f32, interval.1 as f32, 0, std::ptr::null(), &mut hit) };
        assert!(st == 0, "rt_intersect_batch: {}", ffi::last_error());
        if hit.hit == 0 {
            return None;
        }
        let d = |v: [f32; 3]| [v[0] as f64, v[1] as f64, v[2] as f64];
        Some(IntersectionPoint {
            point: Point3(d(hit.point)),
            normal: UnitVec3::assume_unit(Vec3(d(hit.normal))),
            t: hit.t as f64,
            in_vec: UnitVec3::assume_unit(Vec3(d(hit.in_vec))),
            surface_coordinates: (hit.u as f64, hit.v as f64),
        })
    }
}

/// What is known about the point where a ray meets a shape.
#[derive(Debug, Clone, Copy)]
pub struct IntersectionPoint {
    pub point: Point3,
    pub normal: UnitVec3,
    pub t: f64,
    pub in_vec: UnitVec3,
    pub surface_coordinates: (f64, f64),
}

impl IntersectionPoint {
    pub fn face(&self) -> Face {
        if self.in_vec.dot(*self.normal) < 0.0 {
            Face::Front
        } else {
            Face::Back
        }
    }
}

#[derive(Debug, Clone, Copy)]
pub enum Face {
    Front,
    Back,
}

/// An axis-aligned bounding box.
#[derive(Debug, Clone, Copy)]
pub struct BoundingBox {
    pub min: Point3,
    pub max: Point3,
}

impl BoundingBox {
    /// Whether `ray` meets the box within the open parameter interval (the slab test with true divisions; a slab whose
    /// interval closes to a point counts as missed).
    pub fn hit(&self, ray: &Ray, (mut tmin, mut tmax): (f64, f64)) -> bool {
        for i in 0..3 {
            let (d, o) = (ray.direction[i], ray.origin[i]);
            let (near, far) = ((self.min[i] - o) / d, (self.max[i] - o) / d);
            let (near, far) = if d < 0.0 { (far, near) } else { (near, far) };
            tmin = near.max(tmin);
            tmax = far.min(tmax);
            if tmax <= tmin {
                return false;
            }
        }
        true
    }
}

/// The smallest box that holds both.
impl Add<BoundingBox> for BoundingBox {
    type Output = Self;
    fn add(self, other: Self) -> Self {
        let mut out = self;
        for c in 0..3 {
            out.min[c] = self.min[c].min(other.min[c]);
            out.max[c] = self.max[c].max(other.max[c]);
        }
        out
    }
}
