//! The textures of the public API (the reference's `src/textures.rs`): same enum, same variants and fields.
//! `color_at` is answered on the host (it is a public helper of the crate, not the render path); the device evaluates
//! textures from the flattened tables `crate::flatten` writes.
use super::geometry::{Point3, Vec3};
use image::RgbImage;
use noise::NoiseFn;
use std::fmt;
use std::sync::{Arc, Mutex};

#[derive(Clone)]
pub enum Texture {
    SolidColor(Vec3),
    Checkered { odd: Box<Texture>, even: Box<Texture> },
    Noise { scale: f64, noise: Arc<dyn NoiseFn<[f64; 3]> + Sync + Send> },
    Image(RgbImage),
}

impl fmt::Debug for Texture {
    fn fmt(&self, f: &mut fmt::Formatter<'_>) -> fmt::Result {
        match self {
            Texture::SolidColor(c) => write!(f, "SolidColor({:?})", c),
            Texture::Checkered { odd, even } => write!(f, "Checkered {{ odd: {:?}, even: {:?} }}", odd, even),
            Texture::Noise { scale, .. } => write!(f, "Noise {{ scale: {:?}, noise: <noise function> }}", scale),
            Texture::Image(im) => write!(f, "Image({}x{})", im.width(), im.height()),
        }
    }
}

impl Texture {
    /// Loads an image file as a texture.  Panics if the file cannot be read or is not 8-bit RGB.
    pub fn image<P: AsRef<std::path::Path>>(path: P) -> Self {
        Texture::Image(image::open(path).unwrap().as_rgb8().unwrap().to_owned())
    }

    /// The texture's colour at surface coordinates `(u, v)` and world point `p`.
    pub fn color_at(&self, u: f64, v: f64, p: &Point3) -> Vec3 {
        match self {
            Texture::SolidColor(c) => *c,
            Texture::Checkered { odd, even } => {
                let s: f64 = p.iter().map(|x| (10.0 * x).sin()).product();
                if s < 0.0 {
                    odd.color_at(u, v, p)
                } else {
                    even.color_at(u, v, p)
                }
            }
            Texture::Noise { noise, scale } => {
                let n = noise.get([p[0] * scale, p[1] * scale, p[2] * scale]);
                Vec3([n, n, n])
            }
            Texture::Image(im) => {
                let (w, h) = (im.width(), im.height());
                let i = ((u.max(0.0).min(1.0) * w as f64) as u32).min(w - 1);
                let j = (((1.0 - v.max(0.0).min(1.0)) * h as f64) as u32).min(h - 1);
                let px = im.get_pixel(i, j);
                Vec3([px[0] as f64 / 255.0, px[1] as f64 / 255.0, px[2] as f64 / 255.0])
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// What the device can evaluate of an opaque `dyn NoiseFn`.
//
// The device implements the gradient noise of `noise::Perlin` and two functions built on it (`noise::Fbm`,
// `noise::Turbulence<Perlin>`: see `identify_noise` below).  The gradient noise: `noise::Perlin` (eight surflets attn^4 * (d . grad),
// attn = 1 - |d|^2, scaled by 3.8898553255531074 and clamped to [-1, 1]) over a 256-entry permutation table that crosses
// the C ABI.  A `dyn NoiseFn` does not say what it is, so the facade *identifies* it: for every candidate seed it builds
// the table the way `noise` 0.6 builds `PermutationTable::new(seed)`, evaluates the formula above on the host in f64 at a
// fixed set of probe points and compares with `noise.get()`.  The first table that reproduces the function to 1e-9 is
// handed to the device; if none does the scene is refused (there is no silent substitute).  The table construction and
// the formula are restatements of the crate: this check is what pins them the first time the facade runs next to the
// real crate.

static CANDIDATE_SEEDS: Mutex<Vec<u32>> = Mutex::new(Vec::new());

/// Tells the facade that scenes may use `noise::Perlin::new().set_seed(seed)` (seed 0, the default, is always tried).
pub fn register_perlin_seed(seed: u32) {
    let mut seeds = CANDIDATE_SEEDS.lock().unwrap();
    if !seeds.contains(&seed) {
        seeds.push(seed);
    }
}

/// The permutation table of `noise::Perlin` for `seed`: 0..=255 shuffled (`rand` 0.7 `SliceRandom::shuffle`) by an
/// `XorShiftRng` seeded with the bytes `[1, 0, 0, 0, seed, seed, seed]` as four little-endian words.
pub(crate) fn perlin_table(seed: u32) -> [u8; 256] {
    let mut s = [1u32, seed, seed, seed];
    let mut next = move || {
        let t = s[0] ^ (s[0] << 11);
        s[0] = s[1];
        s[1] = s[2];
        s[2] = s[3];
        s[3] = s[3] ^ (s[3] >> 19) ^ (t ^ (t >> 8));
        s[3]
    };
    let mut table = [0u8; 256];
    for (i, t) in table.iter_mut().enumerate() {
        *t = i as u8;
    }
    for i in (1..256u32).rev() {
        // gen_range(0, i + 1) of rand 0.7 for u32: widening multiply, reject the low word above the zone
        let range = i + 1;
        let zone = (range << range.leading_zeros()).wrapping_sub(1);
        let idx = loop {
            let m = (next() as u64) * (range as u64);
            if (m as u32) <= zone {
                break (m >> 32) as usize;
            }
        };
        table.swap(i as usize, idx);
    }
    table
}

fn gradient(index: u8) -> [f64; 3] {
    let d = std::f64::consts::FRAC_1_SQRT_2;
    let e = 1.0 / 3f64.sqrt();
    let mut i = (index & 31) as usize;
    let sign = |bit: usize, x: f64| if i & bit != 0 { -x } else { x };
    if i >= 24 {
        return [sign(1, e), sign(2, e), sign(4, e)];
    }
    if i >= 12 {
        i -= 12;
    }
    let (a, b) = (if i & 1 != 0 { -d } else { d }, if i & 2 != 0 { -d } else { d });
    if i < 4 {
        [a, b, 0.0]
    } else if i < 8 {
        [a, 0.0, b]
    } else {
        [0.0, a, b]
    }
}

/// The device's noise function on the host, in f64.
pub(crate) fn device_perlin(table: &[u8; 256], p: [f64; 3]) -> f64 {
    let f = [p[0].floor(), p[1].floor(), p[2].floor()];
    let cell = [f[0] as i64, f[1] as i64, f[2] as i64];
    let mut total = 0.0;
    for corner in 0..8usize {
        let c = [corner & 1, (corner >> 1) & 1, corner >> 2];
        let d = [p[0] - f[0] - c[0] as f64, p[1] - f[1] - c[1] as f64, p[2] - f[2] - c[2] as f64];
        let attn = 1.0 - (d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        if attn > 0.0 {
            let at = |k: usize| ((cell[k] + c[k] as i64) & 255) as usize;
            let h = table[table[table[at(0)] as usize ^ at(1)] as usize ^ at(2)];
            let g = gradient(h);
            total += attn.powi(4) * (d[0] * g[0] + d[1] * g[1] + d[2] * g[2]);
        }
    }
    (total * 3.889_855_325_553_107_4).max(-1.0).min(1.0)
}

/// The device's `RT_NOISE_FBM` on the host, in f64: octave `i` is the Perlin of seed `seed + i` at the point scaled by
/// `lacunarity^i`, weighted `persistence^i` (`fbm3` in the kernels' rt_core.h).
pub(crate) fn device_fbm(seed: u32, octaves: u32, lacunarity: f64, persistence: f64, p: [f64; 3]) -> f64 {
    let (mut sum, mut weight, mut q) = (0.0, 1.0, p);
    for i in 0..octaves {
        sum += weight * device_perlin(&perlin_table(seed.wrapping_add(i)), q);
        weight *= persistence;
        q = [q[0] * lacunarity, q[1] * lacunarity, q[2] * lacunarity];
    }
    sum
}

/// The device's `RT_NOISE_TURBULENCE` on the host: the source Perlin at the point displaced, per coordinate, by `power` times
/// an Fbm of seeds `seed`, `seed + 1`, `seed + 2` looked up at the point shifted by the crate's fixed fractions of 65536
/// (`turbulence3` in rt_core.h).
pub(crate) fn device_turbulence(source: &[u8; 256], seed: u32, roughness: u32, lacunarity: f64, persistence: f64, power: f64, p: [f64; 3]) -> f64 {
    let u = 1.0 / 65536.0;
    let shifts = [[12414.0, 65124.0, 31337.0], [26519.0, 18128.0, 60493.0], [53820.0, 11213.0, 44845.0]];
    let mut q = p;
    for c in 0..3 {
        let at = [p[0] + shifts[c][0] * u, p[1] + shifts[c][1] * u, p[2] + shifts[c][2] * u];
        q[c] = p[c] + power * device_fbm(seed.wrapping_add(c as u32), roughness, lacunarity, persistence, at);
    }
    device_perlin(source, q)
}

/// What the device evaluates in place of an opaque `dyn NoiseFn`.
pub(crate) enum DeviceNoise {
    Perlin { table: [u8; 256] },
    Fbm { seed: u32, octaves: u32, lacunarity: f64, persistence: f64 },
    Turbulence { source: [u8; 256], seed: u32, roughness: u32, lacunarity: f64, persistence: f64, power: f64 },
}

/// The lacunarity `noise::Fbm` starts with (and the distortion functions of `noise::Turbulence` with it).
pub(crate) const NOISE_CRATE_LACUNARITY: f64 = std::f64::consts::PI * 2.0 / 3.0;

fn probe_points() -> Vec<[f64; 3]> {
    // 64 points spread over many lattice cells, off the lattice
    (0..64u32)
        .map(|k| {
            let k = k as f64;
            [0.37 + 1.618 * k, -5.21 + 2.414 * k, 11.03 - 0.732 * k]
        })
        .collect()
}

fn candidate_seeds() -> Vec<u32> {
    let mut seeds = vec![0u32];
    seeds.extend(CANDIDATE_SEEDS.lock().unwrap().iter().copied());
    seeds
}

/// The permutation table under which the device's noise equals `noise`, if there is one among the candidates.
pub(crate) fn identify_perlin(noise: &(dyn NoiseFn<[f64; 3]> + Sync + Send)) -> Option<[u8; 256]> {
    let probes = probe_points();
    candidate_seeds().into_iter().map(perlin_table).find(|table| probes.iter().all(|p| (device_perlin(table, *p) - noise.get(*p)).abs() < 1e-9))
}

/// Which of the device's noise functions `noise` is, by the same probe-and-compare step: a Perlin of a candidate seed, else a
/// `noise::Fbm` with the crate's starting lacunarity and persistence and 1 to 8 octaves, else a `noise::Turbulence<Perlin>`
/// with the crate's starting power and a roughness of 1 to 4, each over the candidate seeds (source and distortion seeds
/// alike).  Other settings are not guessed: the scene is refused rather than drawn with a stand-in.
pub(crate) fn identify_noise(noise: &(dyn NoiseFn<[f64; 3]> + Sync + Send)) -> Option<DeviceNoise> {
    if let Some(table) = identify_perlin(noise) {
        return Some(DeviceNoise::Perlin { table });
    }
    let probes = probe_points();
    let want: Vec<f64> = probes.iter().map(|p| noise.get(*p)).collect();
    let same = |f: &dyn Fn([f64; 3]) -> f64| probes.iter().zip(&want).all(|(p, w)| (f(*p) - w).abs() < 1e-9);
    let (lacunarity, persistence, power) = (NOISE_CRATE_LACUNARITY, 0.5, 1.0);
    for seed in candidate_seeds() {
        for octaves in (1..=8u32).rev() {
            if same(&|p| device_fbm(seed, octaves, lacunarity, persistence, p)) {
                return Some(DeviceNoise::Fbm { seed, octaves, lacunarity, persistence });
            }
        }
    }
    for source_seed in candidate_seeds() {
        let source = perlin_table(source_seed);
        for seed in candidate_seeds() {
            for roughness in 1..=4u32 {
                if same(&|p| device_turbulence(&source, seed, roughness, lacunarity, persistence, power, p)) {
                    return Some(DeviceNoise::Turbulence { source, seed, roughness, lacunarity, persistence, power });
                }
            }
        }
    }
    None
}
