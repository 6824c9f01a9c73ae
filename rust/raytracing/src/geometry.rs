//! Points, vectors, rays and the samplers of the crate's public API (the reference's `src/geometry.rs`).  These are host
//! types: scene descriptions are written with them and flattened for the device (`crate::flatten`); nothing here is on the
//! render path, which runs in librt_hip.so.
mod shape;

use rand::distributions::{Distribution, Uniform};
use rand::Rng;
use std::iter::Sum;
use std::ops::{Add, AddAssign, Deref, DerefMut, Div, DivAssign, Mul, MulAssign, Neg, Range, Sub, SubAssign};

pub use shape::{Boundable, BoundingBox, Face, Intersection, IntersectionPoint, Shape};

/// A coordinate axis.
#[derive(Debug, Clone, Copy, PartialEq)]
pub enum Axis {
    X,
    Y,
    Z,
}

impl Axis {
    /// The two coordinates a rotation about this axis mixes, in the order (c0, c1) with
    /// `out[c0] = s[c0] cos - s[c1] sin`, `out[c1] = s[c0] sin + s[c1] cos`.
    pub(crate) fn plane(self) -> (usize, usize) {
        match self {
            Axis::X => (1, 2),
            Axis::Y => (2, 0),
            Axis::Z => (0, 1),
        }
    }

    /// (p1, p2, o): the two in-plane coordinates of a rectangle orthogonal to this axis, and the axis itself.
    pub(crate) fn rect_frame(self) -> (usize, usize, usize) {
        match self {
            Axis::X => (1, 2, 0),
            Axis::Y => (2, 0, 1),
            Axis::Z => (0, 1, 2),
        }
    }

    pub(crate) fn index(self) -> u32 {
        self as u32
    }
}

pub(crate) trait Rotate {
    fn rotate(self, axis: Axis, angle: f64) -> Self;
}

fn rotate3(s: [f64; 3], axis: Axis, degrees: f64) -> [f64; 3] {
    let (c0, c1) = axis.plane();
    let (sin, cos) = degrees.to_radians().sin_cos();
    let mut out = s;
    out[c0] = s[c0] * cos - s[c1] * sin;
    out[c1] = s[c0] * sin + s[c1] * cos;
    out
}

macro_rules! triple {
    ($name:ident) => {
        #[derive(Clone, Copy, Debug, PartialEq)]
        pub struct $name(pub [f64; 3]);

        // (a slice, as in the reference, geometry.rs:52-64 and :131-143: `p[i]`, `p.iter()`, `p.len()` mean the same there and here)
        impl Deref for $name {
            type Target = [f64];
            fn deref(&self) -> &[f64] {
                &self.0
            }
        }
        impl DerefMut for $name {
            fn deref_mut(&mut self) -> &mut [f64] {
                &mut self.0
            }
        }
        impl Default for $name {
            fn default() -> Self {
                $name([0.0; 3])
            }
        }
        impl Rotate for $name {
            fn rotate(self, axis: Axis, angle: f64) -> Self {
                $name(rotate3(self.0, axis, angle))
            }
        }
    };
}
triple!(Point3);
triple!(Vec3);

fn zip3(a: [f64; 3], b: [f64; 3], f: impl Fn(f64, f64) -> f64) -> [f64; 3] {
    [f(a[0], b[0]), f(a[1], b[1]), f(a[2], b[2])]
}

// componentwise operators: $lhs op $rhs -> $out
macro_rules! binop {
    ($tr:ident, $f:ident, $lhs:ident, $rhs:ident, $out:ident, $op:tt) => {
        impl $tr<$rhs> for $lhs {
            type Output = $out;
            fn $f(self, o: $rhs) -> $out {
                $out(zip3(self.0, o.0, |a, b| a $op b))
            }
        }
    };
}
macro_rules! assignop {
    ($tr:ident, $f:ident, $lhs:ident, $rhs:ident, $op:tt) => {
        impl $tr<$rhs> for $lhs {
            fn $f(&mut self, o: $rhs) {
                self.0 = zip3(self.0, o.0, |a, b| a $op b);
            }
        }
    };
}
binop!(Add, add, Vec3, Vec3, Vec3, +);
binop!(Add, add, Point3, Vec3, Point3, +);
binop!(Sub, sub, Vec3, Vec3, Vec3, -);
binop!(Sub, sub, Point3, Vec3, Point3, -);
binop!(Sub, sub, Point3, Point3, Vec3, -);
binop!(Mul, mul, Vec3, Vec3, Vec3, *);
assignop!(AddAssign, add_assign, Vec3, Vec3, +);
assignop!(AddAssign, add_assign, Point3, Vec3, +);
assignop!(SubAssign, sub_assign, Vec3, Vec3, -);
assignop!(SubAssign, sub_assign, Point3, Vec3, -);
assignop!(MulAssign, mul_assign, Vec3, Vec3, *);

impl Mul<f64> for Vec3 {
    type Output = Vec3;
    fn mul(self, s: f64) -> Vec3 {
        Vec3([self[0] * s, self[1] * s, self[2] * s])
    }
}
impl MulAssign<f64> for Vec3 {
    fn mul_assign(&mut self, s: f64) {
        *self = *self * s;
    }
}
impl Div<f64> for Vec3 {
    type Output = Vec3;
    fn div(self, s: f64) -> Vec3 {
        Vec3([self[0] / s, self[1] / s, self[2] / s])
    }
}
impl DivAssign<f64> for Vec3 {
    fn div_assign(&mut self, s: f64) {
        *self = *self / s;
    }
}
impl Neg for Vec3 {
    type Output = Vec3;
    fn neg(self) -> Vec3 {
        Vec3([-self[0], -self[1], -self[2]])
    }
}
impl Sum<Vec3> for Vec3 {
    fn sum<I: Iterator<Item = Vec3>>(iter: I) -> Vec3 {
        iter.fold(Vec3::default(), |acc, v| acc + v)
    }
}

impl Point3 {
    /// Euclidean distance to another point.
    pub fn dist(&self, other: &Point3) -> f64 {
        (*self - *other).norm()
    }

    /// Three independent uniform draws from `range`.  (Not in the reference crate, whose `examples/final_scene.rs`
    /// nevertheless calls it; with it that example needs only its moved `sphere` value cloned to build.)
    pub fn random<R: Rng + ?Sized>(range: Range<f64>, rng: &mut R) -> Self {
        Point3(Vec3::random(range, rng).0)
    }
}

impl Vec3 {
    pub fn norm(&self) -> f64 {
        self.norm_squared().sqrt()
    }

    pub fn norm_squared(&self) -> f64 {
        self.dot(*self)
    }

    pub fn dot(self, other: Self) -> f64 {
        self[0] * other[0] + self[1] * other[1] + self[2] * other[2]
    }

    pub fn cross(self, other: Self) -> Self {
        Vec3([
            self[1] * other[2] - self[2] * other[1],
            self[2] * other[0] - self[0] * other[2],
            self[0] * other[1] - self[1] * other[0],
        ])
    }

    /// The vector scaled to length one (NaN components for the zero vector, as a division by the norm gives).
    pub fn normed(&self) -> UnitVec3 {
        UnitVec3(*self / self.norm())
    }

    /// Three independent uniform draws from `range`.
    pub fn random<R: Rng + ?Sized>(range: Range<f64>, rng: &mut R) -> Self {
        let d = Uniform::from(range);
        Vec3([d.sample(rng), d.sample(rng), d.sample(rng)])
    }
}

/// A vector of length one.
#[derive(Copy, Clone, Debug)]
pub struct UnitVec3(Vec3);

impl UnitVec3 {
    /// Wraps a vector that is already of length one (records coming back from the device).
    pub(crate) fn assume_unit(v: Vec3) -> Self {
        UnitVec3(v)
    }

    pub fn rotate(self, axis: Axis, angle: f64) -> Self {
        UnitVec3(Rotate::rotate(self.0, axis, angle))
    }
}
impl Deref for UnitVec3 {
    type Target = Vec3;
    fn deref(&self) -> &Vec3 {
        &self.0
    }
}
impl Neg for UnitVec3 {
    type Output = UnitVec3;
    fn neg(self) -> UnitVec3 {
        UnitVec3(-self.0)
    }
}

/// An orthonormal basis.
#[derive(Copy, Clone, Debug)]
pub struct Onb([UnitVec3; 3]);

impl Onb {
    /// The basis whose third vector points along `w` and whose second lies in the plane of `v` and `w`.
    pub fn from23(v: Vec3, w: Vec3) -> Self {
        let e2 = w.normed();
        let e0 = v.cross(*e2).normed();
        let e1 = e2.cross(*e0).normed();
        Onb([e0, e1, e2])
    }
}
impl Deref for Onb {
    type Target = [UnitVec3; 3];
    fn deref(&self) -> &[UnitVec3; 3] {
        &self.0
    }
}

/// A ray: all points `origin + t * direction`.
#[derive(Clone, Copy, Debug)]
pub struct Ray {
    pub origin: Point3,
    pub direction: Vec3,
}

impl Ray {
    /// The point at parameter `t`; the origin for `t <= 0`.
    pub fn at(&self, t: f64) -> Point3 {
        if t <= 0.0 {
            self.origin
        } else {
            self.origin + self.direction * t
        }
    }
}

/// Uniform distribution inside the unit ball (rejection from the cube).
pub struct InsideUnitSphere;

impl Distribution<Vec3> for InsideUnitSphere {
    fn sample<R: Rng + ?Sized>(&self, rng: &mut R) -> Vec3 {
        loop {
            let v = Vec3::random(-1.0..1.0, rng);
            if v.norm() < 1.0 {
                return v;
            }
        }
    }
}
impl Distribution<Point3> for InsideUnitSphere {
    fn sample<R: Rng + ?Sized>(&self, rng: &mut R) -> Point3 {
        let v: Vec3 = self.sample(rng);
        Point3(v.0)
    }
}

/// The crate's direction sampler: longitude and height uniform, radius `1 - z^2` (no square root), then normalised.
pub struct OnUnitSphere;

impl Distribution<UnitVec3> for OnUnitSphere {
    fn sample<R: Rng + ?Sized>(&self, rng: &mut R) -> UnitVec3 {
        let phi = rng.gen_range(0.0, 2.0 * std::f64::consts::PI);
        let z: f64 = rng.gen_range(-1.0, 1.0);
        let r = 1.0 - z * z;
        Vec3([r * phi.cos(), r * phi.sin(), z]).normed()
    }
}
impl Distribution<Point3> for OnUnitSphere {
    fn sample<R: Rng + ?Sized>(&self, rng: &mut R) -> Point3 {
        let v: UnitVec3 = self.sample(rng);
        Point3((*v).0)
    }
}

pub fn random_unit_vector<R: Rng + ?Sized>(rng: &mut R) -> UnitVec3 {
    OnUnitSphere.sample(rng)
}

/// Uniform distribution inside the unit disc of the xy plane (rejection from the square).
pub struct InsideUnitDisc;

impl Distribution<Vec3> for InsideUnitDisc {
    fn sample<R: Rng + ?Sized>(&self, rng: &mut R) -> Vec3 {
        let d = Uniform::from(-1.0..1.0);
        loop {
            let v = Vec3([d.sample(rng), d.sample(rng), 0.0]);
            if v.norm() < 1.0 {
                return v;
            }
        }
    }
}
