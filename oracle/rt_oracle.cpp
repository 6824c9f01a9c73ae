// rt_oracle.cpp -- CPU restatement (f64) of the reference path tracer's hot path.
//
// TEST INFRASTRUCTURE ONLY.  Nothing in the shipped library links, loads or
// calls this file; only tests/, __graft_entry__.smoke() and the cpu_baseline
// leg of bench.py may use it, and only as the checker / reported baseline.
//
// What it restates (all paths relative to the reference checkout):
//   src/lib.rs:18-104      pixels / pixel / ray_color
//   src/lib.rs:153-257     BvhNode::create_ / scatter_
//   src/lib.rs:276-288     Object::scatter
//   src/lib.rs:343-402     Camera::new / ray / random_time
//   src/geometry.rs        Vec3 ops, rotate, Onb::from23, Ray::at, samplers
//   src/geometry/shape.rs  Shape::intersect / bound, BoundingBox::hit
//   src/materials.rs       Material::scatter, reflect, refract, schlick
//   src/textures.rs        Texture::color_at
//   examples/*.rs          the five scene definitions
//
// Pinning status: the reference is Rust and cannot be built in this
// environment; its own tests hold no numeric golden vectors (two `is_some()`
// intersection tests and two sampler property tests, src/geometry.rs:410-466).
// The oracle is pinned against exactly those; beyond them PARITY IS UNPINNED
// and rests on line-by-line review against the cited lines.
//
// Third-party arithmetic that is not in the reference tree:
//   noise 0.6.0 (Cargo.lock:325) `Perlin::get` and its seed-0 permutation
//   table are restated from the crate's published algorithm from memory
//   (surflet gradient noise) -- unverified, parity unpinned.
//   rand 0.7.3: only the distributions matter; every random draw here comes
//   from a counter-based Philox4x32-10 keyed on (pixel, sample, bounce, stream)
//   so that the GPU path can consume identical variates.
//
// Deliberate deviations from the literal reference (all distribution-neutral):
//   * rand::thread_rng() -> keyed Philox (the reference is OS-seeded and not
//     reproducible even against itself);
//   * Object::scatter shades every accepted BVH candidate (lib.rs:287); with
//     keyed draws only the final closest candidate's shading is observable, so
//     shading of discarded candidates is done only when `shade_every_candidate`
//     is set (used for the timed CPU baseline);
//   * rayon's tree reduction of sample sums -> sequential sum in sample order.

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <string>
#include <thread>
#include <vector>

namespace orc {

static const double PI = 3.14159265358979323846264338327950288;  // std::f64::consts::PI
static const double INF = std::numeric_limits<double>::infinity();

// ---------------------------------------------------------------- counters
// Work counters on the reference traversal order; they feed the algorithmic
// flop figure used as the roofline numerator.
struct Counters {
    uint64_t samples = 0, segments = 0;
    uint64_t slab = 0;
    uint64_t sphere_miss = 0, sphere_range = 0, sphere_hit = 0;
    uint64_t rect_treject = 0, rect_breject = 0, rect_hit = 0;
    uint64_t box_tests = 0, translate = 0, rotate_miss = 0, rotate_hit = 0;
    uint64_t medium = 0, moving = 0;
    uint64_t lambertian = 0, metal = 0, dielectric = 0, light = 0, isotropic = 0;
    uint64_t tex_checker = 0, tex_noise = 0, tex_image = 0;
    void add(const Counters& o) {
        const uint64_t* s = reinterpret_cast<const uint64_t*>(&o);
        uint64_t* d = reinterpret_cast<uint64_t*>(this);
        for (size_t i = 0; i < sizeof(Counters) / sizeof(uint64_t); ++i) d[i] += s[i];
    }
};
static thread_local Counters* g_cnt = nullptr;
#define CNT(field) do { if (g_cnt) g_cnt->field++; } while (0)

// ------------------------------------------------------------------- Vec3
// geometry.rs:25, :74 -- Point3 and Vec3 are both [f64;3]; one type here.
struct Vec3 {
    double v[3];
    double& operator[](int i) { return v[i]; }
    double operator[](int i) const { return v[i]; }
};
static inline Vec3 V(double x, double y, double z) { return Vec3{{x, y, z}}; }
static inline Vec3 operator+(Vec3 a, Vec3 b) { return V(a[0] + b[0], a[1] + b[1], a[2] + b[2]); }  // geometry.rs:151
static inline Vec3 operator-(Vec3 a, Vec3 b) { return V(a[0] - b[0], a[1] - b[1], a[2] - b[2]); }  // geometry.rs:190
static inline Vec3 operator-(Vec3 a) { return V(-a[0], -a[1], -a[2]); }                            // geometry.rs:183
static inline Vec3 operator*(Vec3 a, double s) { return V(a[0] * s, a[1] * s, a[2] * s); }         // geometry.rs:230
static inline Vec3 operator*(Vec3 a, Vec3 b) { return V(a[0] * b[0], a[1] * b[1], a[2] * b[2]); }  // geometry.rs:245
static inline Vec3 operator/(Vec3 a, double s) { return V(a[0] / s, a[1] / s, a[2] / s); }         // geometry.rs:260
static inline double norm_squared(Vec3 a) { return a[0] * a[0] + a[1] * a[1] + a[2] * a[2]; }      // geometry.rs:83 (powi(2))
static inline double norm(Vec3 a) { return std::sqrt(norm_squared(a)); }                           // geometry.rs:78
static inline double dot(Vec3 a, Vec3 b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }       // geometry.rs:88
static inline Vec3 cross(Vec3 a, Vec3 b) {                                                         // geometry.rs:93
    return V(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]);
}
static inline Vec3 normed(Vec3 a) { return a / norm(a); }  // geometry.rs:102 (NaN for the zero vector)

// geometry.rs:33-50 and :112-129 -- rotation about a coordinate axis, angle in degrees.
enum Axis { AX = 0, AY = 1, AZ = 2 };
static inline Vec3 rotate(Vec3 s, int axis, double angle_deg) {
    Vec3 out = s;
    double angle = angle_deg * (PI / 180.0);  // f64::to_radians
    double c = std::cos(angle), sn = std::sin(angle);
    int c0, c1;
    switch (axis) {
        case AX: c0 = 1; c1 = 2; break;
        case AY: c0 = 2; c1 = 0; break;
        default: c0 = 0; c1 = 1; break;
    }
    out[c0] = s[c0] * c - s[c1] * sn;
    out[c1] = s[c0] * sn + s[c1] * c;
    return out;
}

// geometry.rs:333-347
struct Ray {
    Vec3 origin, direction;
    bool unclamped = false;  // NOT the reference: at() without its clamp (set only inside the boundary queries of a multi-scatter medium)
    Vec3 at(double t) const {
        if (t <= 0.0 && !unclamped) return origin;  // the reference clamps: t <= 0 returns the origin
        return origin + direction * t;
    }
};

// f64::max / f64::min ignore a NaN operand, as fmax/fmin do.
static inline double rmax(double a, double b) { return std::fmax(a, b); }
static inline double rmin(double a, double b) { return std::fmin(a, b); }

// ----------------------------------------------------------------- Philox
// Philox4x32-10 (Salmon et al. 2011).  counter = (pixel, sample, bounce_tag,
// stream), key = the 64-bit render seed.
struct U4 { uint32_t x, y, z, w; };
static inline U4 philox4x32_10(U4 c, uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)M0 * c.x, p1 = (uint64_t)M1 * c.z;
        U4 n;
        n.x = (uint32_t)(p1 >> 32) ^ c.y ^ k0;
        n.y = (uint32_t)p1;
        n.z = (uint32_t)(p0 >> 32) ^ c.w ^ k1;
        n.w = (uint32_t)p0;
        c = n;
        k0 += W0; k1 += W1;
    }
    return c;
}
// Uniform variate in the open interval (0,1): the 2^23 odd multiples of 2^-24, exactly representable in fp32
// and f64 alike (so the GPU path can consume identical variates).  Never 0 or 1: with rand's 53-bit variates
// the reference all but never draws the endpoints either (z = -1 in OnUnitSphere gives the exact pole, and
// "random_unit_vector + normal" = 0 on a z-facing surface turns the whole pixel NaN).
static inline double u01(uint32_t x) { return (double)((x >> 9) * 2u + 1u) * (1.0 / 16777216.0); }

static const uint32_t TAG_CAMERA = 0xFFFFFFFFu;

struct DrawCtx {
    uint64_t seed;
    uint32_t pixel, sample, bounce;
    const double* medium_u_override;  // if non-null, *medium_u_override is used by every medium
    U4 draw(uint32_t tag, uint32_t stream) const {
        return philox4x32_10(U4{pixel, sample, tag, stream}, (uint32_t)seed, (uint32_t)(seed >> 32));
    }
    double medium_u(uint32_t prim) const {
        if (medium_u_override) return *medium_u_override;
        return u01(draw(bounce, 1u + prim).x);
    }
};

// geometry.rs:372-382 -- OnUnitSphere: phi~U[0,2pi), z~U[-1,1), r = 1 - z*z (no sqrt), normalised.
static inline Vec3 random_unit_vector(double u_phi, double u_z) {
    double phi = 0.0 + (2.0 * PI - 0.0) * u_phi;
    double z = -1.0 + (1.0 - -1.0) * u_z;
    double r = 1.0 - z * z;
    return normed(V(r * std::cos(phi), r * std::sin(phi), z));
}

// geometry.rs:397-408 -- InsideUnitDisc by rejection; candidate k comes from the
// camera stream: round 1+k/2 of the Philox counter holds two (x,y) candidates.
static inline Vec3 inside_unit_disc(const DrawCtx& ctx) {
    for (uint32_t round = 1;; ++round) {
        U4 r = ctx.draw(TAG_CAMERA, round);
        Vec3 p = V(-1.0 + 2.0 * u01(r.x), -1.0 + 2.0 * u01(r.y), 0.0);
        if (!(norm(p) >= 1.0)) return p;
        p = V(-1.0 + 2.0 * u01(r.z), -1.0 + 2.0 * u01(r.w), 0.0);
        if (!(norm(p) >= 1.0)) return p;
    }
}

// ------------------------------------------------------------ BoundingBox
// shape.rs:571-625
struct BoundingBox {
    Vec3 min, max;
    bool hit(const Ray& ray, double tmin, double tmax) const {  // shape.rs:580-599
        CNT(slab);
        for (int i = 0; i < 3; ++i) {
            double d = ray.direction[i], o = ray.origin[i];
            double t0 = (min[i] - o) / d, t1 = (max[i] - o) / d;
            if (d < 0.0) std::swap(t0, t1);
            tmin = rmax(t0, tmin);
            tmax = rmin(t1, tmax);
            if (tmax <= tmin) return false;
        }
        return true;
    }
};
static inline BoundingBox operator+(const BoundingBox& a, const BoundingBox& b) {  // shape.rs:604-625
    return BoundingBox{V(rmin(a.min[0], b.min[0]), rmin(a.min[1], b.min[1]), rmin(a.min[2], b.min[2])),
                       V(rmax(a.max[0], b.max[0]), rmax(a.max[1], b.max[1]), rmax(a.max[2], b.max[2]))};
}

// shape.rs:539-567
struct IntersectionPoint {
    Vec3 point, normal;
    double t;
    Vec3 in_vec;
    double u, v;
    bool front_face() const { return dot(in_vec, normal) < 0.0; }  // shape.rs:554
};

// ------------------------------------------------------------------ Shape
// shape.rs:62-106 -- the closed enum of eight variants, kept as a recursive tree.
struct Shape;
typedef std::shared_ptr<const Shape> ShapeP;
struct Shape {
    enum Kind { Sphere, MovingSphere, Rectangle, ConstantMedium, Box, Flipped, Translated, Rotated } kind;
    // Sphere / MovingSphere
    Vec3 center{}; double radius = 0; Ray ray{}; double start_time = 0, end_time = 0;
    bool both_roots = false;  // NOT the reference: try the far root when the near one is out of range (a variant for comparison)
    bool multi_scatter = false;  // NOT the reference (ConstantMedium): boundary queries with Ray::at unclamped, so that a ray starting inside scatters again
    // Rectangle
    int axis = 0; double x0 = 0, y0 = 0, x1 = 0, y1 = 0, height = 0;
    // Box
    Vec3 bmin{}, bmax{}; std::vector<ShapeP> sides;
    // wrappers
    ShapeP inner; Vec3 offset{}; double angle = 0; double density = 0;

    bool intersect(const Ray& ray, double tmin, double tmax, double time, const DrawCtx& ctx, uint32_t prim,
                   IntersectionPoint* out) const;
    BoundingBox bound() const;
};

static void axis_perm(int axis, int& p1, int& p2, int& o) {  // shape.rs:138-142
    switch (axis) {
        case AX: p1 = 1; p2 = 2; o = 0; break;
        case AY: p1 = 2; p2 = 0; o = 1; break;
        default: p1 = 0; p2 = 1; o = 2; break;
    }
}

static ShapeP make_sphere(Vec3 c, double r) {
    auto s = std::make_shared<Shape>(); s->kind = Shape::Sphere; s->center = c; s->radius = r; return s;
}
// shape.rs:222-245 -- Shape::sphere; returns null if end_time <= start_time (the reference asserts).
static ShapeP make_sphere_timed(double radius, double t0, Vec3 c0, double t1, Vec3 c1) {
    if (!(t1 > t0)) return nullptr;
    if (c0[0] == c1[0] && c0[1] == c1[1] && c0[2] == c1[2]) return make_sphere(c0, radius);
    auto s = std::make_shared<Shape>();
    s->kind = Shape::MovingSphere; s->start_time = t0; s->end_time = t1; s->radius = radius;
    s->ray = Ray{c0, c1 - c0};
    return s;
}
static ShapeP make_rect_raw(int axis, double x0, double y0, double x1, double y1, double height) {
    auto s = std::make_shared<Shape>();
    s->kind = Shape::Rectangle; s->axis = axis; s->x0 = x0; s->y0 = y0; s->x1 = x1; s->y1 = y1; s->height = height;
    return s;
}
// shape.rs:129-154
static ShapeP make_rectangle(Vec3 ll, Vec3 ur) {
    int axis = (ll[0] == ur[0]) ? AX : (ll[1] == ur[1]) ? AY : AZ;
    int p1, p2, o; axis_perm(axis, p1, p2, o);
    return make_rect_raw(axis, ll[p1], ll[p2], ur[p1], ur[p2], ll[o]);
}
static ShapeP make_flipped(ShapeP in) {
    auto s = std::make_shared<Shape>(); s->kind = Shape::Flipped; s->inner = in; return s;
}
static ShapeP make_translated(ShapeP in, Vec3 off) {
    auto s = std::make_shared<Shape>(); s->kind = Shape::Translated; s->inner = in; s->offset = off; return s;
}
static ShapeP make_rotated(ShapeP in, int axis, double angle) {
    auto s = std::make_shared<Shape>(); s->kind = Shape::Rotated; s->inner = in; s->axis = axis; s->angle = angle; return s;
}
static ShapeP make_medium(ShapeP boundary, double density) {
    auto s = std::make_shared<Shape>(); s->kind = Shape::ConstantMedium; s->inner = boundary; s->density = density; return s;
}
// shape.rs:157-215 -- six rectangles x0(F) x1 y0(F) y1 z0(F) z1.
static ShapeP make_box(Vec3 mn, Vec3 mx) {
    auto s = std::make_shared<Shape>();
    s->kind = Shape::Box; s->bmin = mn; s->bmax = mx;
    double X0 = mn[0], Y0 = mn[1], Z0 = mn[2], X1 = mx[0], Y1 = mx[1], Z1 = mx[2];
    s->sides.push_back(make_flipped(make_rect_raw(AX, Y0, Z0, Y1, Z1, X0)));
    s->sides.push_back(make_rect_raw(AX, Y0, Z0, Y1, Z1, X1));
    s->sides.push_back(make_flipped(make_rect_raw(AY, X0, Z0, X1, Z1, Y0)));
    s->sides.push_back(make_rect_raw(AY, X0, Z0, X1, Z1, Y1));
    s->sides.push_back(make_flipped(make_rect_raw(AZ, X0, Y0, X1, Y1, Z0)));
    s->sides.push_back(make_rect_raw(AZ, X0, Y0, X1, Y1, Z1));
    return s;
}

// shape.rs:248-426
bool Shape::intersect(const Ray& ray, double tmin, double tmax, double time, const DrawCtx& ctx, uint32_t prim,
                      IntersectionPoint* out) const {
    switch (kind) {
        case Sphere: {  // shape.rs:256-293
            Vec3 oc = ray.origin - center;
            double a = norm_squared(ray.direction);
            double half_b = dot(oc, ray.direction);
            double c = norm_squared(oc) - radius * radius;
            double discriminant = half_b * half_b - a * c;
            if (discriminant < 0.0) { CNT(sphere_miss); return false; }
            double t = (-half_b - std::sqrt(discriminant)) / a;  // near root only
            if (both_roots && !(t > tmin && t < tmax)) t = (-half_b + std::sqrt(discriminant)) / a;  // (the variant: then the far one)
            if (t > tmin && t < tmax) {
                CNT(sphere_hit);
                Vec3 point = ray.at(t);
                Vec3 normal = normed(point - center);
                Vec3 p = (point - center) / radius;
                double phi = std::atan2(p[2], p[0]);
                double theta = std::asin(p[1]);
                out->u = 1.0 - (phi + PI) / (2.0 * PI);
                out->v = (theta + PI / 2.0) / PI;
                out->t = t; out->point = point; out->normal = normal; out->in_vec = normed(ray.direction);
                return true;
            }
            CNT(sphere_range);
            return false;
        }
        case MovingSphere: {  // shape.rs:295-307
            CNT(moving);
            Shape s; s.kind = Sphere; s.center = this->ray.at(time - start_time); s.radius = radius; s.both_roots = both_roots;
            return s.intersect(ray, tmin, tmax, time, ctx, prim, out);
        }
        case Rectangle: {  // shape.rs:309-344
            int p1, p2, o; axis_perm(axis, p1, p2, o);
            double t = (height - ray.origin[o]) / ray.direction[o];
            if (t < tmin || t > tmax) { CNT(rect_treject); return false; }
            Vec3 point = ray.at(t);
            if (point[p1] < x0 || point[p1] > x1 || point[p2] < y0 || point[p2] > y1) { CNT(rect_breject); return false; }
            CNT(rect_hit);
            out->u = (point[p1] - x0) / (x1 - x0);
            out->v = (point[p2] - y0) / (y1 - y0);
            Vec3 n = V(0, 0, 0); n[o] = 1.0;
            out->point = point; out->normal = normed(n); out->t = t; out->in_vec = normed(ray.direction);
            return true;
        }
        case Flipped: {  // shape.rs:346-353
            if (!inner->intersect(ray, tmin, tmax, time, ctx, prim, out)) return false;
            out->normal = -out->normal;
            return true;
        }
        case Box: {  // shape.rs:355 -> Vec<I>::intersect, shape.rs:17-34
            CNT(box_tests);
            bool any = false;
            IntersectionPoint ip;
            for (const ShapeP& s : sides) {
                if (s->intersect(ray, tmin, tmax, time, ctx, prim, &ip)) { tmax = ip.t; *out = ip; any = true; }
            }
            return any;
        }
        case Translated: {  // shape.rs:356-368
            CNT(translate);
            Ray moved{ray.origin - offset, ray.direction, ray.unclamped};
            if (!inner->intersect(moved, tmin, tmax, time, ctx, prim, out)) return false;
            out->point = out->point + offset;
            return true;
        }
        case Rotated: {  // shape.rs:370-384
            Ray rr{rotate(ray.origin, axis, -angle), rotate(ray.direction, axis, -angle), ray.unclamped};
            if (!inner->intersect(rr, tmin, tmax, time, ctx, prim, out)) { CNT(rotate_miss); return false; }
            CNT(rotate_hit);
            out->point = rotate(out->point, axis, angle);
            out->normal = rotate(out->normal, axis, angle);
            out->in_vec = rotate(out->in_vec, axis, angle);
            return true;
        }
        case ConstantMedium: {  // shape.rs:386-423
            CNT(medium);
            IntersectionPoint ip1, ip2;
            Ray query = ray;
            query.unclamped = multi_scatter;  // (the variant: the boundary is met behind the origin where it is, not at the origin)
            if (!inner->intersect(query, -INF, INF, time, ctx, prim, &ip1)) return false;
            if (!inner->intersect(query, ip1.t + 0.001, INF, time, ctx, prim, &ip2)) return false;
            ip1.t = rmax(ip1.t, tmin);
            ip2.t = rmin(ip2.t, tmax);
            if (ip1.t >= ip2.t) return false;
            ip1.t = rmax(ip1.t, 0.0);
            double length = norm(ray.direction);
            double distance_inside_boundary = (ip2.t - ip1.t) * length;
            double hit_distance = -1.0 / density * std::log(ctx.medium_u(prim));
            if (hit_distance > distance_inside_boundary) return false;
            double t = ip1.t + hit_distance / length;
            out->point = ray.at(t);
            out->normal = normed(V(1.0, 0.0, 0.0));
            out->t = t; out->in_vec = normed(ray.direction); out->u = 0.0; out->v = 0.0;
            return true;
        }
    }
    return false;
}

// shape.rs:428-524
BoundingBox Shape::bound() const {
    switch (kind) {
        case Sphere: return BoundingBox{center - V(1, 1, 1) * radius, center + V(1, 1, 1) * radius};
        case MovingSphere: {
            Shape a; a.kind = Sphere; a.center = ray.at(0.0); a.radius = radius;
            Shape b; b.kind = Sphere; b.center = ray.at(end_time - start_time); b.radius = radius;
            return a.bound() + b.bound();
        }
        case Rectangle: {
            int p1, p2, o; axis_perm(axis, p1, p2, o);
            Vec3 mn = V(0, 0, 0), mx = V(0, 0, 0);
            mn[p1] = x0; mn[p2] = y0; mn[o] = height - 0.0001;
            mx[p1] = x1; mx[p2] = y1; mx[o] = height + 0.0001;
            return BoundingBox{mn, mx};
        }
        case Flipped: return inner->bound();
        case Box: return BoundingBox{bmin, bmax};
        case Translated: { BoundingBox b = inner->bound(); return BoundingBox{b.min + offset, b.max + offset}; }
        case Rotated: {
            BoundingBox bb = inner->bound();
            Vec3 mn = V(INF, INF, INF), mx = V(-INF, -INF, -INF);
            for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int k = 0; k < 2; ++k) {
                double x = i * bb.max[0] + (1.0 - i) * bb.min[0];
                double y = j * bb.max[1] + (1.0 - j) * bb.min[1];
                double z = k * bb.max[2] + (1.0 - k) * bb.min[2];
                Vec3 tester = rotate(V(x, y, z), axis, angle);
                for (int c = 0; c < 3; ++c) { mn[c] = rmin(mn[c], tester[c]); mx[c] = rmax(mx[c], tester[c]); }
            }
            return BoundingBox{mn, mx};
        }
        case ConstantMedium: return inner->bound();
    }
    return BoundingBox{};
}

// ---------------------------------------------------------------- Perlin
// noise 0.6.0 `Perlin` (third party, restated from memory; parity unpinned).
struct PermTable { uint8_t values[256]; };

static uint32_t xorshift_next(uint32_t s[4]) {  // rand_xorshift 0.2
    uint32_t x = s[0], t = x ^ (x << 11);
    s[0] = s[1]; s[1] = s[2]; s[2] = s[3];
    uint32_t w = s[3];
    s[3] = w ^ (w >> 19) ^ (t ^ (t >> 8));
    return s[3];
}
// noise 0.6.0 PermutationTable::new(seed): XorShiftRng seeded with bytes
// [1,0,0,0, seed, seed, seed], then rand 0.7 `shuffle` of 0..=255.
static PermTable perm_table_from_seed(uint32_t seed) {
    uint32_t s[4] = {1u, seed, seed, seed};
    PermTable pt;
    for (int i = 0; i < 256; ++i) pt.values[i] = (uint8_t)i;
    for (uint32_t i = 255; i >= 1; --i) {
        uint32_t range = i + 1;  // gen_range(0, i+1) via widening multiply with rejection zone
        int lz = __builtin_clz(range);
        uint32_t zone = (range << lz) - 1u;
        uint32_t idx;
        for (;;) {
            uint64_t m = (uint64_t)xorshift_next(s) * range;
            if ((uint32_t)m <= zone) { idx = (uint32_t)(m >> 32); break; }
        }
        std::swap(pt.values[i], pt.values[idx]);
    }
    return pt;
}
static inline void perlin_grad3(unsigned index, double g[3]) {
    const double d = 0.7071067811865476, d2 = 0.5773502691896258;
    static const double T[32][3] = {
        {d, d, 0}, {-d, d, 0}, {d, -d, 0}, {-d, -d, 0}, {d, 0, d}, {-d, 0, d}, {d, 0, -d}, {-d, 0, -d},
        {0, d, d}, {0, -d, d}, {0, d, -d}, {0, -d, -d},
        {d, d, 0}, {-d, d, 0}, {d, -d, 0}, {-d, -d, 0}, {d, 0, d}, {-d, 0, d}, {d, 0, -d}, {-d, 0, -d},
        {0, d, d}, {0, -d, d}, {0, d, -d}, {0, -d, -d},
        {d2, d2, d2}, {-d2, d2, d2}, {d2, -d2, d2}, {-d2, -d2, d2},
        {d2, d2, -d2}, {-d2, d2, -d2}, {d2, -d2, -d2}, {-d2, -d2, -d2}};
    const double* r = T[index % 32];
    g[0] = r[0]; g[1] = r[1]; g[2] = r[2];
}
static double perlin3(const PermTable& pt, double x, double y, double z) {
    double fl[3] = {std::floor(x), std::floor(y), std::floor(z)};
    long nc[3] = {(long)fl[0], (long)fl[1], (long)fl[2]};
    double nd[3] = {x - fl[0], y - fl[1], z - fl[2]};
    double total = 0.0;
    for (int cz = 0; cz < 2; ++cz) for (int cy = 0; cy < 2; ++cy) for (int cx = 0; cx < 2; ++cx) {
        long c[3] = {nc[0] + cx, nc[1] + cy, nc[2] + cz};
        double dd[3] = {nd[0] - cx, nd[1] - cy, nd[2] - cz};
        double attn = 1.0 - (dd[0] * dd[0] + dd[1] * dd[1] + dd[2] * dd[2]);
        if (attn > 0.0) {
            unsigned h = pt.values[(unsigned)(c[0] & 0xff)];
            h = pt.values[h ^ (unsigned)(c[1] & 0xff)];
            h = pt.values[h ^ (unsigned)(c[2] & 0xff)];
            double g[3]; perlin_grad3(h, g);
            double a2 = attn * attn;
            total += a2 * a2 * (dd[0] * g[0] + dd[1] * g[1] + dd[2] * g[2]);
        }
    }
    double r = total * 3.8898553255531074;
    return r < -1.0 ? -1.0 : (r > 1.0 ? 1.0 : r);
}

// noise 0.6.0 `Fbm` and `Turbulence<Perlin>` (third party, restated from memory; parity unpinned; no example of the
// reference uses them -- Texture::Noise takes any NoiseFn, textures.rs:14-17).  Fbm: sum over octaves i of
// persistence^i * Perlin(seed + i)(point * lacunarity^i), frequency 1.  Turbulence: the source at the point displaced per
// coordinate by power * Fbm(seed + c)(point + fixed offset), the distortion functions with `roughness` octaves.
static double fbm3(const PermTable* tables, uint32_t octaves, double lacunarity, double persistence, double x, double y, double z) {
    double result = 0.0, amplitude = 1.0;
    for (uint32_t i = 0; i < octaves; ++i) {
        result += perlin3(tables[i], x, y, z) * amplitude;
        amplitude *= persistence;
        x *= lacunarity; y *= lacunarity; z *= lacunarity;
    }
    return result;
}
static double turbulence3(const PermTable* tables, const PermTable& source, uint32_t roughness, double lacunarity, double persistence, double power,
                          double x, double y, double z) {
    const double u = 1.0 / 65536.0;
    const double dx = fbm3(tables, roughness, lacunarity, persistence, x + 12414.0 * u, y + 65124.0 * u, z + 31337.0 * u);
    const double dy = fbm3(tables + 1, roughness, lacunarity, persistence, x + 26519.0 * u, y + 18128.0 * u, z + 60493.0 * u);
    const double dz = fbm3(tables + 2, roughness, lacunarity, persistence, x + 53820.0 * u, y + 11213.0 * u, z + 44845.0 * u);
    return perlin3(source, x + dx * power, y + dy * power, z + dz * power);
}

// ---------------------------------------------------------------- Texture
// textures.rs:8-19, :46-78
struct Texture;
typedef std::shared_ptr<const Texture> TexP;
struct Texture {
    enum Kind { SolidColor, Checkered, Noise, Image } kind;
    Vec3 color{};
    TexP odd, even;
    double scale = 1.0; PermTable perm{};
    int function = 0;  // 0 Perlin, 1 Fbm, 2 Turbulence (perm: its source)
    uint32_t octaves = 0; double lacunarity = 0.0, persistence = 0.0, power = 0.0; std::vector<PermTable> octave_tables;
    uint32_t width = 0, height = 0; std::vector<uint8_t> rgb;
    Vec3 color_at(double u, double v, const Vec3& p) const {
        switch (kind) {
            case SolidColor: return color;
            case Checkered:
                CNT(tex_checker);
                if (std::sin(10.0 * p[0]) * std::sin(10.0 * p[1]) * std::sin(10.0 * p[2]) < 0.0) return odd->color_at(u, v, p);
                return even->color_at(u, v, p);
            case Noise:
                CNT(tex_noise);
                if (function == 1) return V(1.0, 1.0, 1.0) * fbm3(octave_tables.data(), octaves, lacunarity, persistence, p[0] * scale, p[1] * scale, p[2] * scale);
                if (function == 2)
                    return V(1.0, 1.0, 1.0) * turbulence3(octave_tables.data(), perm, octaves, lacunarity, persistence, power, p[0] * scale, p[1] * scale, p[2] * scale);
                return V(1.0, 1.0, 1.0) * perlin3(perm, p[0] * scale, p[1] * scale, p[2] * scale);
            case Image: {
                CNT(tex_image);
                u = rmin(rmax(u, 0.0), 1.0);
                v = 1.0 - rmin(rmax(v, 0.0), 1.0);
                uint32_t i = (uint32_t)(u * (double)width);   // `as u32` saturates; u in [0,1] so no overflow
                uint32_t j = (uint32_t)(v * (double)height);
                i = std::min(i, width - 1); j = std::min(j, height - 1);
                const uint8_t* px = &rgb[3 * ((size_t)j * width + i)];
                return V(px[0] / 255.0, px[1] / 255.0, px[2] / 255.0);
            }
        }
        return V(0, 0, 0);
    }
};

// --------------------------------------------------------------- Material
// materials.rs:8-148
struct RayHit { bool scattered; Ray ray; Vec3 color; };  // Scattered{ray, attenuation=color} | Emitted(color)

static inline Vec3 reflect(Vec3 v, Vec3 n) { return normed(v - n * 2.0 * dot(v, n)); }  // materials.rs:132
static inline Vec3 refract(Vec3 v, Vec3 n, double eta_ratio) {                           // materials.rs:136
    double cos_theta = -dot(v, n);
    Vec3 out_par = (v + n * cos_theta) * eta_ratio;
    Vec3 out_perp = n * -(std::sqrt(1.0 - norm_squared(out_par)));
    return out_par + out_perp;
}
static inline double schlick(double cosine, double refraction_index) {  // materials.rs:143
    double r0 = (1.0 - refraction_index) / (1.0 + refraction_index);
    r0 = r0 * r0;
    double m = 1.0 - cosine;
    return r0 + (1.0 - r0) * (m * m * m * m * m);
}

struct Material {
    enum Kind { Lambertian, Metal, Dielectric, DiffuseLight, Isotropic } kind;
    TexP texture; Vec3 albedo{}; double param = 0;  // fuzz | refraction_index
    bool face_normal = false;  // NOT the reference: the material sees the normal turned against the incoming ray (a variant for comparison)
    RayHit scatter(const IntersectionPoint& p_in, const DrawCtx& ctx) const {
        U4 r = ctx.draw(ctx.bounce, 0);
        const bool front = p_in.front_face();
        IntersectionPoint p = p_in;
        if (face_normal && !front) p.normal = -p.normal;
        switch (kind) {
            case Lambertian: {  // materials.rs:30-47, :116-129
                CNT(lambertian);
                Vec3 direction = random_unit_vector(u01(r.x), u01(r.y)) + p.normal;
                Ray scattered{p.point, direction};
                Vec3 sd = normed(scattered.direction);
                double cosine = dot(p.normal, sd);
                double pdf = rmax(cosine / PI, 0.0);
                Vec3 att = texture->color_at(p.u, p.v, p.point) * pdf * PI / dot(p.normal, normed(scattered.direction));
                return RayHit{true, scattered, att};
            }
            case Metal: {  // materials.rs:49-63
                CNT(metal);
                Vec3 reflected = reflect(p.in_vec, p.normal);
                Vec3 scattered = reflected + random_unit_vector(u01(r.x), u01(r.y)) * param;
                if (dot(scattered, p.normal) >= 0.0) return RayHit{true, Ray{p.point, scattered}, albedo};
                return RayHit{false, Ray{}, V(0, 0, 0)};
            }
            case Dielectric: {  // materials.rs:65-93
                CNT(dielectric);
                double eta_ratio = front ? 1.0 / param : param;
                Vec3 in_unit = normed(p.in_vec);
                double cos_theta = -rmin(dot(in_unit, p.normal), 1.0);
                double sin_theta = std::sqrt(1.0 - cos_theta * cos_theta);
                Vec3 direction;
                if (eta_ratio * sin_theta > 1.0 || u01(r.z) < schlick(cos_theta, eta_ratio)) direction = reflect(in_unit, p.normal);
                else direction = refract(in_unit, p.normal, eta_ratio);
                return RayHit{true, Ray{p.point, direction}, albedo};
            }
            case DiffuseLight:  // materials.rs:95-99
                CNT(light);
                return RayHit{false, Ray{}, texture->color_at(p.u, p.v, p.point)};
            case Isotropic:  // materials.rs:101-112
                CNT(isotropic);
                return RayHit{true, Ray{p.point, random_unit_vector(u01(r.x), u01(r.y))}, texture->color_at(p.u, p.v, p.point)};
        }
        return RayHit{false, Ray{}, V(0, 0, 0)};
    }
};
typedef std::shared_ptr<const Material> MatP;

// ------------------------------------------------------ Object, BVH, Scene
struct Object { ShapeP shape; MatP material; uint32_t id; };  // lib.rs:261-266 (+ index in the input list)

// Seeded generator for scene parameters and BVH split axes (the reference uses
// the OS-seeded thread_rng): SplitMix64; uniform double = top 53 bits * 2^-53.
struct SceneRng {
    uint64_t s;
    explicit SceneRng(uint64_t seed) : s(seed) {}
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
    double range(double lo, double hi) { return lo + (hi - lo) * uniform(); }
    uint32_t below(uint32_t n) { return (uint32_t)(uniform() * n); }
};

struct HitResult { double t; RayHit rh; IntersectionPoint ip; uint32_t prim; };

struct BvhNode {  // lib.rs:137-151
    bool leaf; BoundingBox bb; Object object; std::unique_ptr<BvhNode> left, right;

    // lib.rs:160-200
    static std::unique_ptr<BvhNode> create(std::vector<Object>& objs, size_t lo, size_t hi, SceneRng& rng) {
        size_t n = hi - lo;
        auto node = std::make_unique<BvhNode>();
        if (n == 1) {
            node->leaf = true; node->object = objs[lo]; node->bb = objs[lo].shape->bound();
            return node;
        }
        int axis = (int)rng.below(3);  // drawn before recursing (pre-order)
        std::stable_sort(objs.begin() + lo, objs.begin() + hi, [axis](const Object& a, const Object& b) {
            return a.shape->bound().min[axis] < b.shape->bound().min[axis];
        });
        node->leaf = false;
        node->left = create(objs, lo, lo + n / 2, rng);
        node->right = create(objs, lo + n / 2, hi, rng);
        node->bb = node->left->bb + node->right->bb;
        return node;
    }

    // lib.rs:228-257 -- left first, always; right with tmax = t_left; a right hit replaces.
    bool scatter_(const Ray& ray, double tmin, double tmax, double time, DrawCtx& ctx, bool shade_every, HitResult* out) const {
        if (!bb.hit(ray, tmin, tmax)) return false;
        if (leaf) {  // Object::scatter, lib.rs:276-288
            IntersectionPoint ip;
            if (!object.shape->intersect(ray, tmin, tmax, time, ctx, object.id, &ip)) return false;
            out->t = ip.t; out->ip = ip; out->prim = object.id;
            if (shade_every) out->rh = object.material->scatter(ip, ctx);
            return true;
        }
        bool any = false;
        HitResult h;
        if (left->scatter_(ray, tmin, tmax, time, ctx, shade_every, &h)) { *out = h; tmax = h.t; any = true; }
        if (right->scatter_(ray, tmin, tmax, time, ctx, shade_every, &h)) { *out = h; any = true; }
        return any;
    }
    void collect_leaf_boxes(std::vector<BoundingBox>& out) const {
        if (leaf) { if (out.size() <= object.id) out.resize(object.id + 1); out[object.id] = bb; }
        else { left->collect_leaf_boxes(out); right->collect_leaf_boxes(out); }
    }
    void leaf_order(std::vector<uint32_t>& out) const {
        if (leaf) out.push_back(object.id); else { left->leaf_order(out); right->leaf_order(out); }
    }
    void count(size_t& nodes, size_t& depth, size_t d) const {
        nodes++; depth = std::max(depth, d);
        if (!leaf) { left->count(nodes, depth, d + 1); right->count(nodes, depth, d + 1); }
    }
};

struct Scene {  // lib.rs:120-133
    std::vector<Object> objects;  // input order (prim ids)
    std::unique_ptr<BvhNode> root;
    Vec3 background;
    bool hit(const Ray& ray, double tmin, double tmax, double time, DrawCtx& ctx, bool shade_every, HitResult* out) const {
        if (!root->scatter_(ray, tmin, tmax, time, ctx, shade_every, out)) return false;
        if (!shade_every) out->rh = objects[out->prim].material->scatter(out->ip, ctx);
        return true;
    }
    // Order-independent statement of what the BVH computes: an object can be hit only by a ray that
    // overlaps the object's own bounding box somewhere in (tmin, tmax) (the leaf test of lib.rs:236-240
    // with the query's original tmax); among those, the closest accepted hit wins.
    bool hit_gated(const Ray& ray, double tmin, double tmax, double time, DrawCtx& ctx, HitResult* out) const {
        std::vector<BoundingBox> boxes;
        root->collect_leaf_boxes(boxes);
        const double tmax0 = tmax;
        bool any = false;
        IntersectionPoint ip;
        Counters* saved = g_cnt; g_cnt = nullptr;
        for (const Object& o : objects) {
            if (!boxes[o.id].hit(ray, tmin, tmax0)) continue;
            if (o.shape->intersect(ray, tmin, tmax, time, ctx, o.id, &ip)) { tmax = ip.t; out->t = ip.t; out->ip = ip; out->prim = o.id; any = true; }
        }
        g_cnt = saved;
        return any;
    }
    // closest accepted hit by brute force over the object list (cross-check of the BVH)
    bool hit_brute(const Ray& ray, double tmin, double tmax, double time, DrawCtx& ctx, HitResult* out) const {
        bool any = false;
        IntersectionPoint ip;
        for (const Object& o : objects)
            if (o.shape->intersect(ray, tmin, tmax, time, ctx, o.id, &ip)) { tmax = ip.t; out->t = ip.t; out->ip = ip; out->prim = o.id; any = true; }
        return any;
    }
};

// ----------------------------------------------------------------- Camera
// lib.rs:314-402
struct CameraOptions { double aperture, aspect_ratio, focus_distance; Vec3 looking_at, origin; double shutter_close, shutter_open, vfov; Vec3 vup; };
struct Camera {
    Vec3 horizontal, vertical, lower_left_corner, origin, onb[3];
    double lens_radius, shutter_open, shutter_close;
    explicit Camera(const CameraOptions& o) {  // lib.rs:347-384
        double h = std::tan((o.vfov / 2.0) * (PI / 180.0));
        double viewport_height = 2.0 * h;
        double viewport_width = viewport_height * o.aspect_ratio;
        Vec3 w = normed(o.origin - o.looking_at);  // Onb::from23, geometry.rs:313-319
        Vec3 u = normed(cross(o.vup, w));
        Vec3 v = normed(cross(w, u));
        onb[0] = u; onb[1] = v; onb[2] = w;
        horizontal = u * viewport_width * o.focus_distance;
        vertical = v * viewport_height * o.focus_distance;
        lower_left_corner = o.origin - horizontal / 2.0 - vertical / 2.0 - w * o.focus_distance;
        lens_radius = o.aperture / 2.0;
        origin = o.origin; shutter_open = o.shutter_open; shutter_close = o.shutter_close;
    }
    Ray ray(double s, double t, const DrawCtx& ctx) const {  // lib.rs:389-396
        Vec3 rd = inside_unit_disc(ctx) * lens_radius;
        Vec3 offset = onb[0] * rd[0] + onb[1] * rd[1];
        Vec3 org = origin + offset;
        Vec3 direction = (lower_left_corner + horizontal * s + vertical * t) - org;
        return Ray{org, direction};
    }
    double random_time(double u) const { return shutter_open + (shutter_close - shutter_open) * u; }  // lib.rs:399
};

// ---------------------------------------------------------------- render
// lib.rs:82-104
// Analysis switch, never set by tests or the bench: with ORC_F32_ORIGINS in the environment a ray's origin is rounded to fp32
// before every segment, as a renderer that keeps its rays in fp32 does, to show what that rounding alone does to an image
// (ORC_F32_ORIGINS=<mask>: only origins on the primitives whose bit is set; cornell with 0x20, its tall box: -0.3 % in the
// frame mean, from rays along the box's sides that meet the side they left).
static unsigned long f32_origins() {
    static const unsigned long mask = std::getenv("ORC_F32_ORIGINS") ? std::strtoul(std::getenv("ORC_F32_ORIGINS"), nullptr, 0) : 0ul;
    return mask;
}
static Vec3 ray_color(Ray ray, const Scene& scene, double time, DrawCtx& ctx, size_t max_reflections, bool shade_every) {
    Vec3 result = V(1.0, 1.0, 1.0);
    long last_prim = -1;
    for (size_t b = 0; b < max_reflections; ++b) {
        if (last_prim >= 0 && ((f32_origins() >> (last_prim & 63)) & 1ul))
            for (int c = 0; c < 3; ++c) ray.origin[c] = (double)(float)ray.origin[c];
        ctx.bounce = (uint32_t)b;
        CNT(segments);
        HitResult h;
        if (!scene.hit(ray, 0.001, INF, time, ctx, shade_every, &h)) return result * scene.background;
        if (!h.rh.scattered) return result * h.rh.color;
        result = result * h.rh.color;
        ray = h.rh.ray;
        last_prim = (long)h.prim;
    }
    return V(0.0, 0.0, 0.0);
}

// lib.rs:40-73 -- returns the mean radiance of pixel (i, j); i = column, j = row counted from the bottom.
static Vec3 pixel_mean(const Camera& cam, const Scene& scene, uint32_t i, uint32_t j, uint32_t width, uint32_t height,
                       size_t spp, size_t max_reflections, uint64_t seed, bool shade_every) {
    Vec3 sum = V(0, 0, 0);
    for (size_t s = 0; s < spp; ++s) {
        DrawCtx ctx{seed, j * width + i, (uint32_t)s, 0, nullptr};
        U4 r = ctx.draw(TAG_CAMERA, 0);
        double u = ((double)i + u01(r.x)) / (double)(width - 1);
        double v = ((double)j + u01(r.y)) / (double)(height - 1);
        CNT(samples);
        Ray ray = cam.ray(u, v, ctx);
        double time = cam.random_time(u01(r.z));
        sum = sum + ray_color(ray, scene, time, ctx, max_reflections, shade_every);
    }
    return sum / (double)spp;
}
static inline uint8_t quantise(double x) {  // lib.rs:68-72; `as u8` saturates and maps NaN to 0
    double y = std::sqrt(rmin(rmax(x, 0.0), 1.0)) * 255.999;
    if (!(y > 0.0)) return 0;
    if (y >= 255.0) return 255;
    return (uint8_t)y;
}

// --------------------------------------------------------- example scenes
// The five scene definitions of examples/*.rs with every thread_rng draw
// replaced by SceneRng(scene_seed) in the same order as the source.
struct Example { std::vector<ShapeP> shapes; std::vector<MatP> mats; Vec3 background; CameraOptions cam; };

static TexP tex_solid(Vec3 c) { auto t = std::make_shared<Texture>(); t->kind = Texture::SolidColor; t->color = c; return t; }
static TexP tex_checker(TexP odd, TexP even) { auto t = std::make_shared<Texture>(); t->kind = Texture::Checkered; t->odd = odd; t->even = even; return t; }
static TexP tex_noise(double scale, const PermTable& pt) { auto t = std::make_shared<Texture>(); t->kind = Texture::Noise; t->scale = scale; t->perm = pt; return t; }
static TexP tex_image(uint32_t w, uint32_t h, const uint8_t* rgb) {
    auto t = std::make_shared<Texture>(); t->kind = Texture::Image; t->width = w; t->height = h; t->rgb.assign(rgb, rgb + (size_t)3 * w * h); return t;
}
static MatP mat_make(Material::Kind k, TexP tex, Vec3 albedo, double param) {
    auto m = std::make_shared<Material>(); m->kind = k; m->texture = tex; m->albedo = albedo; m->param = param; return m;
}
static MatP mat_lambertian(TexP t) { return mat_make(Material::Lambertian, t, V(0, 0, 0), 0); }
static MatP mat_metal(Vec3 a, double fuzz) { return mat_make(Material::Metal, nullptr, a, fuzz); }
static MatP mat_dielectric(Vec3 a, double ri) { return mat_make(Material::Dielectric, nullptr, a, ri); }
static MatP mat_glass() { return mat_dielectric(V(1, 1, 1), 1.5); }  // materials.rs:21-26
static MatP mat_light(TexP t) { return mat_make(Material::DiffuseLight, t, V(0, 0, 0), 0); }
static MatP mat_isotropic(TexP t) { return mat_make(Material::Isotropic, t, V(0, 0, 0), 0); }

static void push(Example& e, ShapeP s, MatP m) { e.shapes.push_back(s); e.mats.push_back(m); }

// Vec3::random (geometry.rs:106-109): three draws in x, y, z order.
static Vec3 vec3_random(double lo, double hi, SceneRng& rng) {
    double x = rng.range(lo, hi);
    double y = rng.range(lo, hi);
    double z = rng.range(lo, hi);
    return V(x, y, z);
}

// examples/random_spheres.rs:23-116
static Example example_random_spheres(SceneRng& rng) {
    Example e;
    e.cam = CameraOptions{0.1, 16.0 / 9.0, 10.0, V(0, 0, 0), V(13, 2, 3), 0.5, 0.0, 20.0, V(0, 1, 0)};
    push(e, make_sphere(V(0, -1000, 0), 1000), mat_lambertian(tex_checker(tex_solid(V(0.9, 0.9, 0.9)), tex_solid(V(0.2, 0.3, 0.1)))));
    for (int a = -11; a < 11; ++a) for (int b = -11; b < 11; ++b) {
        double cx = a + 0.9 * rng.uniform();
        double cz = b + 0.9 * rng.uniform();
        Vec3 center = V(cx, 0.2, cz);
        if (norm(center - V(4.0, 0.2, 0.0)) > 0.9) {
            double roll = rng.uniform();
            Vec3 center2 = center; MatP m;
            if (roll < 0.8) {
                center2 = center + V(0.0, rng.range(0.0, 0.5), 0.0);
                Vec3 c1 = vec3_random(0.0, 1.0, rng);
                Vec3 c2 = vec3_random(0.0, 1.0, rng);
                m = mat_lambertian(tex_solid(c1 + c2));
            } else if (roll < 0.95) {
                Vec3 albedo = vec3_random(0.5, 1.0, rng);
                double fuzz = rng.range(0.0, 0.5);
                m = mat_metal(albedo, fuzz);
            } else {
                m = mat_glass();
            }
            push(e, make_sphere_timed(0.2, 0.0, center, 1.0, center2), m);
        }
    }
    push(e, make_sphere(V(0, 1, 0), 1.0), mat_glass());
    push(e, make_sphere(V(-4, 1, 0), 1.0), mat_lambertian(tex_solid(V(0.4, 0.2, 0.1))));
    push(e, make_sphere(V(4, 1, 0), 1.0), mat_metal(V(0.7, 0.6, 0.5), 0.0));
    e.background = V(1, 1, 1);
    return e;
}

static CameraOptions cornell_camera() {  // cornell.rs:24-34
    return CameraOptions{0.0, 1.0, 10.0, V(278, 278, 0), V(278, 278, -800), 1.0, 0.0, 40.0, V(0, 1, 0)};
}
static void cornell_walls(Example& e) {  // cornell.rs:38-76
    push(e, make_rectangle(V(0, 0, 0), V(0, 555, 555)), mat_lambertian(tex_solid(V(0.65, 0.05, 0.05))));
    push(e, make_flipped(make_rectangle(V(555, 0, 0), V(555, 555, 555))), mat_lambertian(tex_solid(V(0.12, 0.45, 0.15))));
    push(e, make_rectangle(V(0, 0, 0), V(555, 0, 555)), mat_lambertian(tex_solid(V(0.73, 0.73, 0.73))));
    push(e, make_flipped(make_rectangle(V(0, 555, 0), V(555, 555, 555))), mat_lambertian(tex_solid(V(0.73, 0.73, 0.73))));
    push(e, make_flipped(make_rectangle(V(0, 0, 555), V(555, 555, 555))), mat_lambertian(tex_solid(V(0.73, 0.73, 0.73))));
}
static ShapeP cornell_box1() { return make_translated(make_rotated(make_box(V(0, 0, 0), V(165, 330, 165)), AY, 15.0), V(265, 0, 295)); }
static ShapeP cornell_box2() { return make_translated(make_rotated(make_box(V(0, 0, 0), V(165, 165, 165)), AY, -18.0), V(130, 0, 65)); }

// examples/cornell.rs:24-108
static Example example_cornell() {
    Example e; e.cam = cornell_camera();
    cornell_walls(e);
    push(e, cornell_box1(), mat_lambertian(tex_solid(V(0.73, 0.73, 0.73))));
    push(e, cornell_box2(), mat_lambertian(tex_solid(V(0.73, 0.73, 0.73))));
    push(e, make_rectangle(V(213, 554, 227), V(343, 554, 332)), mat_light(tex_solid(V(15, 15, 15))));
    e.background = V(0, 0, 0);
    return e;
}
// examples/cornell_smoke.rs:24-118
static Example example_cornell_smoke() {
    Example e; e.cam = cornell_camera();
    cornell_walls(e);
    push(e, make_medium(cornell_box1(), 0.01), mat_isotropic(tex_solid(V(0, 0, 0))));
    push(e, make_medium(cornell_box2(), 0.01), mat_isotropic(tex_solid(V(1, 1, 1))));
    push(e, make_rectangle(V(113, 554, 127), V(443, 554, 432)), mat_light(tex_solid(V(7, 7, 7))));
    e.background = V(0, 0, 0);
    return e;
}
// examples/perlin.rs:26-85
static Example example_perlin() {
    Example e;
    e.cam = CameraOptions{0.0, 16.0 / 9.0, 10.0, V(0, 0, 0), V(13, 2, 3), 0.01, 0.0, 20.0, V(0, 1, 0)};
    PermTable pt = perm_table_from_seed(0);  // noise::Perlin::new() -> DEFAULT_SEED 0
    push(e, make_sphere(V(0, -1000, 0), 1000), mat_lambertian(tex_noise(1.0, pt)));
    push(e, make_sphere(V(0, 2, 0), 2), mat_lambertian(tex_noise(1.0, pt)));
    push(e, make_sphere(V(0, 7, 0), 2), mat_light(tex_solid(V(4, 4, 4))));
    push(e, make_rectangle(V(3, 1, -2), V(5, 3, -2)), mat_light(tex_solid(V(4, 4, 4))));
    e.background = V(0, 0, 0);
    return e;
}
// examples/final_scene.rs:24-193.  Two source-level errors of that file are
// resolved the obvious way: `sphere` is cloned before its second use (:109/:115)
// and `Point3::random(0.0..165.0)` (:169) is three uniform draws in [0,165).
static Example example_final_scene(SceneRng& rng, uint32_t img_w, uint32_t img_h, const uint8_t* img_rgb) {
    Example e;
    e.cam = CameraOptions{0.1, 1.0, 10.0, V(0, 0, 0), V(13, 2, 3), 0.5, 0.0, 20.0, V(0, 1, 0)};
    for (int i = 0; i < 20; ++i) for (int j = 0; j < 20; ++j) {
        double w = 100.0;
        double x0 = -1000.0 + i * w, y0 = 0.0, z0 = -1000.0 + j * w;
        double x1 = x0 + w, y1 = rng.range(1.0, 101.0), z1 = z0 + w;
        push(e, make_box(V(x0, y0, z0), V(x1, y1, z1)), mat_lambertian(tex_solid(V(0.48, 0.83, 0.53))));
    }
    push(e, make_rectangle(V(123, 554, 147), V(423, 554, 412)), mat_light(tex_solid(V(7, 7, 7))));
    push(e, make_sphere_timed(50.0, 0.0, V(400, 400, 200), 1.0, V(430, 400, 200)), mat_lambertian(tex_solid(V(0.7, 0.3, 0.1))));
    push(e, make_sphere(V(260, 150, 45), 50), mat_glass());
    push(e, make_sphere(V(0, 150, 145), 50), mat_metal(V(0.8, 0.8, 0.9), 10.0));
    ShapeP sphere = make_sphere(V(360, 150, 145), 70);
    push(e, sphere, mat_glass());
    push(e, make_medium(sphere, 0.2), mat_lambertian(tex_solid(V(0.2, 0.4, 0.9))));
    push(e, make_medium(make_sphere(V(0, 0, 0), 5000), 0.0001), mat_lambertian(tex_solid(V(1, 1, 1))));
    push(e, make_sphere(V(400, 200, 400), 100), mat_lambertian(tex_image(img_w, img_h, img_rgb)));
    push(e, make_sphere(V(220, 280, 300), 80), mat_lambertian(tex_noise(0.1, perm_table_from_seed(0))));
    for (int k = 0; k < 1000; ++k) {
        double x = rng.range(0.0, 165.0), y = rng.range(0.0, 165.0), z = rng.range(0.0, 165.0);
        push(e, make_sphere(V(x, y, z), 10.0), mat_lambertian(tex_solid(V(0.73, 0.73, 0.73))));
    }
    push(e, make_sphere(V(0, -1000, 0), 1000), mat_lambertian(tex_checker(tex_solid(V(0.9, 0.9, 0.9)), tex_solid(V(0.2, 0.3, 0.1)))));
    e.background = V(1, 1, 1);
    return e;
}

static Scene* scene_new(const std::vector<ShapeP>& shapes, const std::vector<MatP>& mats, Vec3 background, SceneRng& rng) {
    if (shapes.empty()) return nullptr;  // lib.rs:161 asserts non-empty
    auto sc = new Scene();
    for (size_t i = 0; i < shapes.size(); ++i) sc->objects.push_back(Object{shapes[i], mats[i], (uint32_t)i});
    std::vector<Object> work = sc->objects;
    sc->root = BvhNode::create(work, 0, work.size(), rng);
    sc->background = background;
    return sc;
}

}  // namespace orc

// ======================================================================
// C interface (ctypes).  Handles are heap-allocated shared_ptr / objects.
// ======================================================================
using namespace orc;

extern "C" {

struct orc_hit { int32_t hit; uint32_t prim; double t; double point[3]; double normal[3]; double in_vec[3]; double u, v; };
struct orc_camera_options { double aperture, aspect_ratio, focus_distance, looking_at[3], origin[3], shutter_close, shutter_open, vfov, vup[3]; };

static Vec3 v3(const double* p) { return V(p[0], p[1], p[2]); }

void* orc_sphere(const double* c, double r) { return new ShapeP(make_sphere(v3(c), r)); }
void* orc_sphere_both_roots(const double* c, double r) {
    auto s = std::make_shared<Shape>(); s->kind = Shape::Sphere; s->center = v3(c); s->radius = r; s->both_roots = true;
    return new ShapeP(s);
}
void* orc_sphere_timed(double r, double t0, const double* c0, double t1, const double* c1) {
    ShapeP s = make_sphere_timed(r, t0, v3(c0), t1, v3(c1));
    return s ? new ShapeP(s) : nullptr;
}
void* orc_rectangle(const double* ll, const double* ur) { return new ShapeP(make_rectangle(v3(ll), v3(ur))); }
void* orc_new_box(const double* mn, const double* mx) { return new ShapeP(make_box(v3(mn), v3(mx))); }
void* orc_flipped(void* s) { return new ShapeP(make_flipped(*(ShapeP*)s)); }
void* orc_translated(void* s, const double* off) { return new ShapeP(make_translated(*(ShapeP*)s, v3(off))); }
void* orc_rotated(void* s, int axis, double angle) { return new ShapeP(make_rotated(*(ShapeP*)s, axis, angle)); }
void* orc_constant_medium(void* s, double density) { return new ShapeP(make_medium(*(ShapeP*)s, density)); }
void* orc_constant_medium_multi(void* s, double density) {
    auto m = std::make_shared<Shape>(*make_medium(*(ShapeP*)s, density)); m->multi_scatter = true;
    return new ShapeP(m);
}
void orc_shape_free(void* s) { delete (ShapeP*)s; }

void* orc_tex_solid(const double* c) { return new TexP(tex_solid(v3(c))); }
void* orc_tex_checkered(void* odd, void* even) { return new TexP(tex_checker(*(TexP*)odd, *(TexP*)even)); }
void* orc_tex_noise(double scale, const uint8_t* perm256) { PermTable pt; std::memcpy(pt.values, perm256, 256); return new TexP(tex_noise(scale, pt)); }
static std::vector<PermTable> tables_from_seeds(uint32_t seed, uint32_t count) {
    std::vector<PermTable> out;
    for (uint32_t i = 0; i < count; ++i) out.push_back(perm_table_from_seed(seed + i));
    return out;
}
void* orc_tex_noise_fbm(double scale, uint32_t seed, uint32_t octaves, double lacunarity, double persistence) {
    if (octaves < 1 || octaves > 32) return nullptr;
    auto t = std::make_shared<Texture>(); t->kind = Texture::Noise; t->scale = scale; t->function = 1; t->octaves = octaves;
    t->lacunarity = lacunarity; t->persistence = persistence; t->octave_tables = tables_from_seeds(seed, octaves);
    return new TexP(t);
}
void* orc_tex_noise_turbulence(double scale, const uint8_t* source_perm256, uint32_t seed, uint32_t roughness, double power, double lacunarity, double persistence) {
    if (roughness < 1 || roughness > 32) return nullptr;
    auto t = std::make_shared<Texture>(); t->kind = Texture::Noise; t->scale = scale; t->function = 2; t->octaves = roughness;
    std::memcpy(t->perm.values, source_perm256, 256);
    t->lacunarity = lacunarity; t->persistence = persistence; t->power = power; t->octave_tables = tables_from_seeds(seed, roughness + 2);
    return new TexP(t);
}
void* orc_tex_image(uint32_t w, uint32_t h, const uint8_t* rgb) { return new TexP(tex_image(w, h, rgb)); }
void orc_tex_free(void* t) { delete (TexP*)t; }

void* orc_mat_lambertian(void* t) { return new MatP(mat_lambertian(*(TexP*)t)); }
void* orc_mat_metal(const double* a, double fuzz) { return new MatP(mat_metal(v3(a), fuzz)); }
void* orc_mat_dielectric(const double* a, double ri) { return new MatP(mat_dielectric(v3(a), ri)); }
void* orc_mat_diffuse_light(void* t) { return new MatP(mat_light(*(TexP*)t)); }
void* orc_mat_isotropic(void* t) { return new MatP(mat_isotropic(*(TexP*)t)); }
void* orc_mat_face_normal(void* m) { auto c = std::make_shared<Material>(**(MatP*)m); c->face_normal = true; return new MatP(c); }
void orc_mat_free(void* m) { delete (MatP*)m; }

void orc_perm_table(uint32_t seed, uint8_t* out256) { PermTable pt = perm_table_from_seed(seed); std::memcpy(out256, pt.values, 256); }
double orc_perlin3(const uint8_t* perm256, double x, double y, double z) { PermTable pt; std::memcpy(pt.values, perm256, 256); return perlin3(pt, x, y, z); }

void orc_philox(const uint32_t* ctr, const uint32_t* key, uint32_t* out) {
    U4 r = philox4x32_10(U4{ctr[0], ctr[1], ctr[2], ctr[3]}, key[0], key[1]);
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}
void orc_random_unit_vector(double u_phi, double u_z, double* out) { Vec3 r = random_unit_vector(u_phi, u_z); std::memcpy(out, r.v, 24); }
void orc_inside_unit_disc(uint64_t seed, uint32_t pixel, uint32_t sample, double* out) {
    DrawCtx ctx{seed, pixel, sample, 0, nullptr}; Vec3 r = inside_unit_disc(ctx); std::memcpy(out, r.v, 24);
}
double orc_schlick(double c, double ri) { return schlick(c, ri); }
uint8_t orc_quantise(double x) { return quantise(x); }
void orc_rotate(const double* v, int axis, double angle, double* out) { Vec3 r = rotate(v3(v), axis, angle); std::memcpy(out, r.v, 24); }
void orc_ray_at(const double* o, const double* d, double t, double* out) { Vec3 r = Ray{v3(o), v3(d)}.at(t); std::memcpy(out, r.v, 24); }
int orc_bbox_hit(const double* mn, const double* mx, const double* o, const double* d, double tmin, double tmax) {
    return BoundingBox{v3(mn), v3(mx)}.hit(Ray{v3(o), v3(d)}, tmin, tmax) ? 1 : 0;
}
void orc_shape_bound(void* s, double* mn, double* mx) { BoundingBox b = (*(ShapeP*)s)->bound(); std::memcpy(mn, b.min.v, 24); std::memcpy(mx, b.max.v, 24); }

static void fill_hit(orc_hit* h, bool hit, const IntersectionPoint& ip, uint32_t prim) {
    std::memset(h, 0, sizeof(*h));
    h->hit = hit ? 1 : 0;
    if (!hit) return;
    h->prim = prim; h->t = ip.t; h->u = ip.u; h->v = ip.v;
    std::memcpy(h->point, ip.point.v, 24); std::memcpy(h->normal, ip.normal.v, 24); std::memcpy(h->in_vec, ip.in_vec.v, 24);
}

// Intersection::intersect (shape.rs:14) on n rays; medium_u (optional) is the
// uniform variate every ConstantMedium uses for ray i.
void orc_shape_intersect(void* s, size_t n, const double* o, const double* d, const double* time, double tmin, double tmax,
                         const double* medium_u, orc_hit* out) {
    const Shape& sh = **(ShapeP*)s;
    for (size_t i = 0; i < n; ++i) {
        DrawCtx ctx{0, (uint32_t)i, 0, 0, medium_u ? &medium_u[i] : nullptr};
        IntersectionPoint ip{};
        bool hit = sh.intersect(Ray{v3(o + 3 * i), v3(d + 3 * i)}, tmin, tmax, time[i], ctx, 0, &ip);
        fill_hit(&out[i], hit, ip, 0);
    }
}

void* orc_scene_new(size_t n, void** shapes, void** mats, const double* background, uint64_t bvh_seed) {
    std::vector<ShapeP> s; std::vector<MatP> m;
    for (size_t i = 0; i < n; ++i) { s.push_back(*(ShapeP*)shapes[i]); m.push_back(*(MatP*)mats[i]); }
    SceneRng rng(bvh_seed);
    return scene_new(s, m, v3(background), rng);
}
void orc_scene_free(void* sc) { delete (Scene*)sc; }
size_t orc_scene_num_objects(void* sc) { return ((Scene*)sc)->objects.size(); }
void orc_scene_bvh_stats(void* sc, uint64_t* nodes, uint64_t* depth) {
    size_t n = 0, d = 0; ((Scene*)sc)->root->count(n, d, 1); *nodes = n; *depth = d;
}

// Object ids in the order the BVH's leaves are reached (left to right); out holds num_objects entries.
void orc_scene_leaf_order(void* sc, uint32_t* out) {
    std::vector<uint32_t> v; ((Scene*)sc)->root->leaf_order(v);
    std::memcpy(out, v.data(), v.size() * sizeof(uint32_t));
}

// The five example scenes; `name` in {random_spheres, cornell, perlin, cornell_smoke, final_scene}.
void* orc_example_scene(const char* name, uint64_t scene_seed, uint32_t img_w, uint32_t img_h, const uint8_t* img_rgb,
                        orc_camera_options* cam_out) {
    SceneRng rng(scene_seed);
    Example e;
    std::string nm(name);
    if (nm == "random_spheres") e = example_random_spheres(rng);
    else if (nm == "cornell") e = example_cornell();
    else if (nm == "perlin") e = example_perlin();
    else if (nm == "cornell_smoke") e = example_cornell_smoke();
    else if (nm == "final_scene") { if (!img_rgb) return nullptr; e = example_final_scene(rng, img_w, img_h, img_rgb); }
    else return nullptr;
    if (cam_out) {
        const CameraOptions& c = e.cam;
        cam_out->aperture = c.aperture; cam_out->aspect_ratio = c.aspect_ratio; cam_out->focus_distance = c.focus_distance;
        std::memcpy(cam_out->looking_at, c.looking_at.v, 24); std::memcpy(cam_out->origin, c.origin.v, 24);
        cam_out->shutter_close = c.shutter_close; cam_out->shutter_open = c.shutter_open; cam_out->vfov = c.vfov;
        std::memcpy(cam_out->vup, c.vup.v, 24);
    }
    return scene_new(e.shapes, e.mats, e.background, rng);  // BVH axes continue the same stream, as Scene::new(.., &mut rng) does
}

// Closest hit over the whole scene for n rays at (tmin,tmax): mode 0 = reference
// BVH order (lib.rs:228-257), mode 1 = brute force over the object list.
// Medium variates are Philox(seed; pixel=i, sample=0, bounce=0, stream=1+prim)
// unless medium_u is given.
void orc_scene_intersect(void* scp, size_t n, const double* o, const double* d, const double* time, double tmin, double tmax,
                         uint64_t seed, const double* medium_u, int mode, orc_hit* out) {
    const Scene& sc = *(Scene*)scp;
    for (size_t i = 0; i < n; ++i) {
        DrawCtx ctx{seed, (uint32_t)i, 0, 0, medium_u ? &medium_u[i] : nullptr};
        Ray r{v3(o + 3 * i), v3(d + 3 * i)};
        HitResult h{};
        bool hit;
        if (mode == 0) hit = sc.root->scatter_(r, tmin, tmax, time[i], ctx, false, &h);
        else if (mode == 1) hit = sc.hit_brute(r, tmin, tmax, time[i], ctx, &h);
        else hit = sc.hit_gated(r, tmin, tmax, time[i], ctx, &h);
        fill_hit(&out[i], hit, h.ip, h.prim);
    }
}

// Camera rays exactly as pixel() generates them (lib.rs:55-60): out_o/out_d/out_time per (pixel, sample).
void orc_camera_rays(const orc_camera_options* co, uint32_t width, uint32_t height, uint64_t seed, size_t n,
                     const uint32_t* px_i, const uint32_t* px_j, const uint32_t* sample, double* out_o, double* out_d, double* out_time) {
    CameraOptions c{co->aperture, co->aspect_ratio, co->focus_distance, v3(co->looking_at), v3(co->origin), co->shutter_close,
                    co->shutter_open, co->vfov, v3(co->vup)};
    Camera cam(c);
    for (size_t k = 0; k < n; ++k) {
        DrawCtx ctx{seed, px_j[k] * width + px_i[k], sample[k], 0, nullptr};
        U4 r = ctx.draw(TAG_CAMERA, 0);
        double u = ((double)px_i[k] + u01(r.x)) / (double)(width - 1);
        double v = ((double)px_j[k] + u01(r.y)) / (double)(height - 1);
        Ray ray = cam.ray(u, v, ctx);
        std::memcpy(out_o + 3 * k, ray.origin.v, 24); std::memcpy(out_d + 3 * k, ray.direction.v, 24);
        out_time[k] = cam.random_time(u01(r.z));
    }
}
// The derived camera frame (lib.rs:347-384) for comparison with the host library.
void orc_camera_frame(const orc_camera_options* co, double* out /* origin, lower_left, horizontal, vertical, u, v : 18 doubles, then lens_radius */) {
    CameraOptions c{co->aperture, co->aspect_ratio, co->focus_distance, v3(co->looking_at), v3(co->origin), co->shutter_close,
                    co->shutter_open, co->vfov, v3(co->vup)};
    Camera cam(c);
    const Vec3* vs[6] = {&cam.origin, &cam.lower_left_corner, &cam.horizontal, &cam.vertical, &cam.onb[0], &cam.onb[1]};
    for (int k = 0; k < 6; ++k) std::memcpy(out + 3 * k, vs[k]->v, 24);
    out[18] = cam.lens_radius;
}

// One shading event: Material::scatter on a given intersection point with the
// draws of (seed, pixel, sample, bounce).  out: scattered flag, colour, new ray.
void orc_material_scatter(void* mat, const orc_hit* h, uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t bounce,
                          int32_t* scattered, double* color, double* ray_o, double* ray_d) {
    IntersectionPoint ip{v3(h->point), v3(h->normal), h->t, v3(h->in_vec), h->u, h->v};
    DrawCtx ctx{seed, pixel, sample, bounce, nullptr};
    RayHit rh = (*(MatP*)mat)->scatter(ip, ctx);
    *scattered = rh.scattered ? 1 : 0;
    std::memcpy(color, rh.color.v, 24); std::memcpy(ray_o, rh.ray.origin.v, 24); std::memcpy(ray_d, rh.ray.direction.v, 24);
}
// Same, with the material of object `h->prim` of a scene.
void orc_scene_scatter(void* scp, const orc_hit* h, uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t bounce, int32_t* scattered, double* color,
                       double* ray_o, double* ray_d) {
    const Scene& sc = *(Scene*)scp;
    MatP m = sc.objects[h->prim].material;
    orc_material_scatter(&m, h, seed, pixel, sample, bounce, scattered, color, ray_o, ray_d);
}
void orc_texture_color(void* tex, double u, double v, const double* p, double* out) {
    Vec3 c = (*(TexP*)tex)->color_at(u, v, v3(p)); std::memcpy(out, c.v, 24);
}

// pixels() (lib.rs:18-32) over rows [row_begin,row_end) of the OUTPUT image (row 0 = top),
// `threads` std::threads over interleaved rows.  rgb_out: 3*w*h u8 (top row first);
// linear_out (optional): 3*w*h f64 mean radiance in the same order; counters_out (optional):
// the Counters struct as uint64[].  Returns seconds of wall time spent rendering.
double orc_render(void* scp, const orc_camera_options* co, uint32_t width, uint32_t height, uint64_t spp, uint64_t max_reflections,
                  uint64_t seed, int threads, int shade_every_candidate, uint32_t row_begin, uint32_t row_end, uint8_t* rgb_out,
                  double* linear_out, uint64_t* counters_out) {
    const Scene& sc = *(Scene*)scp;
    CameraOptions c{co->aperture, co->aspect_ratio, co->focus_distance, v3(co->looking_at), v3(co->origin), co->shutter_close,
                    co->shutter_open, co->vfov, v3(co->vup)};
    Camera cam(c);
    if (threads < 1) threads = 1;
    if (row_end > height) row_end = height;
    std::vector<Counters> cnts(threads);
    std::atomic<uint32_t> next_row(row_begin);
    auto t0 = std::chrono::steady_clock::now();
    auto worker = [&](int tid) {
        g_cnt = counters_out ? &cnts[tid] : nullptr;
        for (;;) {
            uint32_t row = next_row.fetch_add(1);
            if (row >= row_end) break;
            uint32_t j = height - 1 - row;  // lib.rs:21-23: rows are emitted for j = height-1 .. 0
            for (uint32_t i = 0; i < width; ++i) {
                Vec3 m = pixel_mean(cam, sc, i, j, width, height, spp, max_reflections, seed, shade_every_candidate != 0);
                size_t k = 3 * ((size_t)row * width + i);
                if (rgb_out) { rgb_out[k] = quantise(m[0]); rgb_out[k + 1] = quantise(m[1]); rgb_out[k + 2] = quantise(m[2]); }
                if (linear_out) { linear_out[k] = m[0]; linear_out[k + 1] = m[1]; linear_out[k + 2] = m[2]; }
            }
        }
        g_cnt = nullptr;
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; ++t) pool.emplace_back(worker, t);
    worker(0);
    for (auto& th : pool) th.join();
    double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (counters_out) {
        Counters total;
        for (auto& cc : cnts) total.add(cc);
        std::memcpy(counters_out, &total, sizeof(Counters));
    }
    return secs;
}
size_t orc_counters_len() { return sizeof(Counters) / sizeof(uint64_t); }
const char* orc_counters_names() {
    return "samples,segments,slab,sphere_miss,sphere_range,sphere_hit,rect_treject,rect_breject,rect_hit,box_tests,translate,"
           "rotate_miss,rotate_hit,medium,moving,lambertian,metal,dielectric,light,isotropic,tex_checker,tex_noise,tex_image";
}

}  // extern "C"
