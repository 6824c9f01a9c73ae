//! From a `Vec<Object>` to the flat arrays of `rt_scene_desc` (include/rt_abi.h): wrappers peeled into per-primitive
//! transform chains, materials and textures into index tables, and the BVH of `Scene::new` as a pre-order array with skip
//! links.  The tree is built exactly as the reference's `BvhNode::create_` builds it (an axis drawn per branch before
//! recursing, stable sort by `bound().min[axis]`, split at `n / 2`): the reference's left-then-right traversal makes the
//! order of the leaves observable, so the device must walk the same tree.
use crate::ffi::*;
use crate::geometry::{Boundable, Shape, Vec3};
use crate::materials::Material;
use crate::textures::{identify_noise, perlin_table, DeviceNoise, Texture};
use crate::Object;

#[derive(Default)]
pub(crate) struct Flat {
    pub prims: Vec<RtPrim>,
    pub spheres: Vec<RtSphere>,
    pub rects: Vec<RtRect>,
    pub boxes: Vec<RtBox>,
    pub xforms: Vec<RtXform>,
    pub materials: Vec<RtMaterial>,
    pub textures: Vec<RtTexture>,
    pub images: Vec<RtImage>,
    pub texels: Vec<u8>,
    pub perm_tables: Vec<u8>,
    pub bvh: Vec<RtBvhNode>,
    pub background: [f64; 3],
}

fn same_bits<T: Copy>(a: &T, b: &T) -> bool {
    let n = std::mem::size_of::<T>();
    // both records were built from `Default` (all padding zero) by field assignment
    unsafe { std::slice::from_raw_parts(a as *const T as *const u8, n) == std::slice::from_raw_parts(b as *const T as *const u8, n) }
}

fn f32x3(v: [f64; 3]) -> [f32; 3] {
    [v[0] as f32, v[1] as f32, v[2] as f32]
}

struct BuildPrim {
    id: u32,
    active: bool,
    exact: bool,
    min: [f64; 3],
    max: [f64; 3],
}

/// The largest fp32 value not above `x - pad` / the smallest not below `x + pad`.
fn round_down(x: f64, pad: f64) -> f32 {
    let mut f = (x - pad) as f32;
    while (f as f64) > x - pad {
        f = step(f, false);
    }
    f
}
fn round_up(x: f64, pad: f64) -> f32 {
    let mut f = (x + pad) as f32;
    while (f as f64) < x + pad {
        f = step(f, true);
    }
    f
}
fn step(f: f32, up: bool) -> f32 {
    if f == 0.0 {
        return if up { f32::from_bits(1) } else { -f32::from_bits(1) };
    }
    let bits = f.to_bits();
    f32::from_bits(if (f > 0.0) == up { bits + 1 } else { bits - 1 })
}

impl Flat {
    /// `axis` supplies the split axis of every branch node, in pre-order (`rng.gen_range(0, 3)` in `Scene::new`).
    /// `keep_inert` keeps primitives that can never be hit (a medium bounded by a sphere) in the tree.
    pub fn from_objects(objects: &[Object], background: Vec3, axis: &mut dyn FnMut() -> usize, keep_inert: bool) -> Flat {
        assert!(!objects.is_empty());
        let mut flat = Flat { background: background.0, ..Flat::default() };
        for object in objects {
            flat.add_object(object);
        }
        let mut build: Vec<BuildPrim> = Vec::with_capacity(objects.len());
        let mut max_abs = 0.0f64;
        for (i, object) in objects.iter().enumerate() {
            let b = object.shape.bound().expect("every shape of the crate has a bounding box");
            let active = keep_inert || flat.prims[i].flags & RT_PRIM_INERT == 0;
            if active {
                for x in b.min.iter().chain(b.max.iter()).filter(|x| x.is_finite()) {
                    max_abs = max_abs.max(x.abs());
                }
            }
            build.push(BuildPrim { id: i as u32, active, exact: flat.prims[i].geom == RT_GEOM_BOX, min: b.min.0, max: b.max.0 });
        }
        // Node boxes are widened by a few fp32 ulp of the scene's largest coordinate: the device's fp32 slab test must
        // never skip a node the reference's f64 test would enter.  Leaves of box-shaped primitives keep their exact box.
        let pad = (4e-6 * max_abs).max(1e-6);
        let mut bvh = Vec::new();
        build_subtree(&mut build, &mut bvh, axis, pad);
        flat.bvh = bvh;
        flat
    }

    /// Index of the first of `tables` in the scene's list of permutation tables, where they stand one behind the other
    /// (appended unless that run of tables is there already).
    fn add_perm_tables(&mut self, tables: &[[u8; 256]]) -> u32 {
        let flat: Vec<u8> = tables.iter().flat_map(|t| t.iter().copied()).collect();
        let have = self.perm_tables.len() / 256;
        for first in 0..(have + 1).saturating_sub(tables.len()) {
            if self.perm_tables[256 * first..256 * (first + tables.len())] == flat[..] {
                return first as u32;
            }
        }
        self.perm_tables.extend_from_slice(&flat);
        have as u32
    }

    fn add_texture(&mut self, texture: &Texture) -> u32 {
        let mut r = RtTexture::default();
        match texture {
            Texture::SolidColor(c) => {
                r.kind = RT_TEX_SOLID;
                r.color = f32x3(c.0);
            }
            Texture::Checkered { odd, even } => {
                r.kind = RT_TEX_CHECKERED;
                r.odd = self.add_texture(odd);
                r.even = self.add_texture(even);
            }
            Texture::Noise { scale, noise } => {
                let found = identify_noise(noise.as_ref()).expect(
                    "Texture::Noise: the device evaluates noise::Perlin, noise::Fbm and noise::Turbulence<Perlin> with the crate's \
                     starting parameters, and this function is none of them for a known seed (textures::register_perlin_seed adds \
                     seeds other than 0)",
                );
                r.kind = RT_TEX_NOISE;
                r.scale = *scale as f32;
                match found {
                    DeviceNoise::Perlin { table } => {
                        r.odd = RT_NOISE_PERLIN;
                        r.resource = self.add_perm_tables(&[table]);
                    }
                    DeviceNoise::Fbm { seed, octaves, lacunarity, persistence } => {
                        // octave i reads the table i places behind the first (rt_abi.h)
                        let tables: Vec<[u8; 256]> = (0..octaves).map(|i| perlin_table(seed.wrapping_add(i))).collect();
                        r.odd = RT_NOISE_FBM | (octaves << 8);
                        r.color = [lacunarity as f32, persistence as f32, 0.0];
                        r.resource = self.add_perm_tables(&tables);
                    }
                    DeviceNoise::Turbulence { source, seed, roughness, lacunarity, persistence, power } => {
                        // the three distortion functions read tables resource + c + i: seeds seed .. seed + roughness + 1 in a row
                        let tables: Vec<[u8; 256]> = (0..roughness + 2).map(|i| perlin_table(seed.wrapping_add(i))).collect();
                        r.odd = RT_NOISE_TURBULENCE | (roughness << 8);
                        r.color = [lacunarity as f32, persistence as f32, power as f32];
                        r.resource = self.add_perm_tables(&tables);
                        r.even = self.add_perm_tables(&[source]);
                    }
                }
            }
            Texture::Image(image) => {
                r.kind = RT_TEX_IMAGE;
                let raw: &[u8] = image.as_raw();
                let known = self.images.iter().position(|im| {
                    im.width == image.width() && im.height == image.height() && &self.texels[im.texel_offset as usize..][..raw.len()] == raw
                });
                r.resource = known.unwrap_or_else(|| {
                    self.images.push(RtImage { width: image.width(), height: image.height(), texel_offset: self.texels.len() as u64 });
                    self.texels.extend_from_slice(raw);
                    self.images.len() - 1
                }) as u32;
            }
        }
        if let Some(k) = self.textures.iter().position(|t| same_bits(t, &r)) {
            return k as u32;
        }
        self.textures.push(r);
        self.textures.len() as u32 - 1
    }

    fn add_object(&mut self, object: &Object) {
        let mut prim = RtPrim { xf_first: self.xforms.len() as u32, ..RtPrim::default() };
        let mut flips = 0u32;
        let mut medium = false;
        let mut shape = &object.shape;
        // peel the wrappers, outermost first
        loop {
            match shape {
                Shape::Flipped(inner) => {
                    if !medium {
                        flips += 1;
                    }
                    shape = inner;
                }
                Shape::Translated { inner, offset } => {
                    self.xforms.push(RtXform { kind: RT_XF_TRANSLATE, a: offset[0], b: offset[1], c: offset[2], ..RtXform::default() });
                    shape = inner;
                }
                Shape::Rotated { axis, inner, angle } => {
                    let (sin, cos) = angle.to_radians().sin_cos();
                    self.xforms.push(RtXform { kind: RT_XF_ROTATE_X + axis.index(), a: cos, b: sin, ..RtXform::default() });
                    shape = inner;
                }
                Shape::ConstantMedium { boundary, density } => {
                    assert!(!medium, "a ConstantMedium nested inside another ConstantMedium is not supported by the device");
                    medium = true;
                    prim.density = *density as f32;
                    prim.xf_outer = self.xforms.len() as u32 - prim.xf_first;
                    shape = boundary;
                }
                _ => break,
            }
        }
        let n_xf = self.xforms.len() - prim.xf_first as usize;
        assert!(n_xf <= RT_MAX_XFORMS_PER_PRIM, "more than {} nested Translated / Rotated wrappers", RT_MAX_XFORMS_PER_PRIM);
        prim.xf_count = n_xf as u16;
        if flips & 1 == 1 {
            prim.flags |= RT_PRIM_FLIP;
        }
        if medium {
            prim.flags |= RT_PRIM_MEDIUM;
        }
        match shape {
            Shape::Sphere { center, radius } => {
                prim.geom = RT_GEOM_SPHERE;
                prim.geom_index = self.spheres.len() as u32;
                self.spheres.push(RtSphere { center: center.0, radius: *radius, ..RtSphere::default() });
            }
            Shape::MovingSphere { ray, radius, start_time, .. } => {
                prim.geom = RT_GEOM_MOVING_SPHERE;
                prim.geom_index = self.spheres.len() as u32;
                self.spheres.push(RtSphere { center: ray.origin.0, radius: *radius, motion: ray.direction.0, start_time: *start_time });
            }
            Shape::Rectangle { axis, lower_left, upper_right, height } => {
                prim.geom = RT_GEOM_RECT;
                prim.geom_index = self.rects.len() as u32;
                self.rects.push(RtRect {
                    axis: axis.index(),
                    reserved: 0,
                    x0: lower_left.0,
                    y0: lower_left.1,
                    x1: upper_right.0,
                    y1: upper_right.1,
                    height: *height,
                });
            }
            Shape::Box { min, max, .. } => {
                prim.geom = RT_GEOM_BOX;
                prim.geom_index = self.boxes.len() as u32;
                self.boxes.push(RtBox { min: min.0, max: max.0 });
            }
            _ => unreachable!(),
        }
        // A medium bounded by a sphere can never be hit: the second boundary query starts past the first hit and
        // recomputes the same near root.
        if medium && prim.geom <= RT_GEOM_MOVING_SPHERE {
            prim.flags |= RT_PRIM_INERT;
        }
        let mut m = RtMaterial::default();
        match &object.material {
            Material::Lambertian { texture } => {
                m.kind = RT_MAT_LAMBERTIAN;
                m.texture = self.add_texture(texture);
            }
            Material::Metal { albedo, fuzz } => {
                m.kind = RT_MAT_METAL;
                m.albedo = f32x3(albedo.0);
                m.param = *fuzz as f32;
            }
            Material::Dielectric { albedo, refraction_index } => {
                m.kind = RT_MAT_DIELECTRIC;
                m.albedo = f32x3(albedo.0);
                m.param = *refraction_index as f32;
            }
            Material::DiffuseLight { emit } => {
                m.kind = RT_MAT_DIFFUSE_LIGHT;
                m.texture = self.add_texture(emit);
            }
            Material::Isotropic { texture } => {
                m.kind = RT_MAT_ISOTROPIC;
                m.texture = self.add_texture(texture);
            }
        }
        prim.material = match self.materials.iter().position(|k| same_bits(k, &m)) {
            Some(k) => k as u32,
            None => {
                self.materials.push(m);
                self.materials.len() as u32 - 1
            }
        };
        self.prims.push(prim);
    }

    /// The description the library copies from; valid while `self` is alive and unchanged.
    pub fn desc(&self) -> RtSceneDesc {
        RtSceneDesc {
            abi_version: RT_ABI_VERSION,
            n_prims: self.prims.len() as u32,
            n_spheres: self.spheres.len() as u32,
            n_rects: self.rects.len() as u32,
            n_boxes: self.boxes.len() as u32,
            n_xforms: self.xforms.len() as u32,
            n_materials: self.materials.len() as u32,
            n_textures: self.textures.len() as u32,
            n_images: self.images.len() as u32,
            n_perm_tables: (self.perm_tables.len() / 256) as u32,
            n_bvh_nodes: self.bvh.len() as u32,
            reserved0: 0,
            prims: self.prims.as_ptr(),
            spheres: self.spheres.as_ptr(),
            rects: self.rects.as_ptr(),
            boxes: self.boxes.as_ptr(),
            xforms: self.xforms.as_ptr(),
            materials: self.materials.as_ptr(),
            textures: self.textures.as_ptr(),
            images: self.images.as_ptr(),
            texels: self.texels.as_ptr(),
            texel_bytes: self.texels.len() as u64,
            perm_tables: self.perm_tables.as_ptr(),
            bvh: self.bvh.as_ptr(),
            background: self.background,
        }
    }
}

/// Emits the subtree over `prims` in pre-order; the union of the active primitives' boxes, or `None` (and nothing
/// emitted) if none of them is active.
fn build_subtree(prims: &mut [BuildPrim], out: &mut Vec<RtBvhNode>, axis: &mut dyn FnMut() -> usize, pad: f64) -> Option<([f64; 3], [f64; 3])> {
    let n = prims.len();
    if n == 1 {
        let p = &prims[0];
        if !p.active {
            return None;
        }
        let mut node = RtBvhNode { prim: p.id | if p.exact { RT_BVH_EXACT } else { 0 }, skip: out.len() as u32 + 1, ..RtBvhNode::default() };
        for c in 0..3 {
            node.min[c] = if p.exact { p.min[c] as f32 } else { round_down(p.min[c], pad) };
            node.max[c] = if p.exact { p.max[c] as f32 } else { round_up(p.max[c], pad) };
        }
        out.push(node);
        return Some((p.min, p.max));
    }
    let k = axis();
    prims.sort_by(|a, b| a.min[k].partial_cmp(&b.min[k]).unwrap_or(std::cmp::Ordering::Equal)); // stable, as the reference's
    let me = out.len();
    out.push(RtBvhNode::default());
    let (left, right) = prims.split_at_mut(n / 2);
    let l = build_subtree(left, out, axis, pad);
    let r = build_subtree(right, out, axis, pad);
    match (l, r) {
        (None, None) => {
            out.pop();
            None
        }
        (Some(one), None) | (None, Some(one)) => {
            // one side is empty: the other side's root takes this node's place
            out.remove(me);
            for node in &mut out[me..] {
                node.skip -= 1;
            }
            Some(one)
        }
        (Some((lmin, lmax)), Some((rmin, rmax))) => {
            let (mut min, mut max) = (lmin, lmax);
            for c in 0..3 {
                min[c] = lmin[c].min(rmin[c]);
                max[c] = lmax[c].max(rmax[c]);
                out[me].min[c] = round_down(min[c], pad);
                out[me].max[c] = round_up(max[c], pad);
            }
            out[me].prim = RT_BVH_NONE;
            out[me].skip = out.len() as u32;
            Some((min, max))
        }
    }
}
