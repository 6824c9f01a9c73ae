// rt_core.h -- per-ray arithmetic of the path tracer: intersection, BVH
// traversal, materials, textures, camera, Philox.  Everything here is a
// function of plain values and read-only scene arrays, marked RT_HD so that the
// HIP kernels (rt_kernels.hip) are its only product consumer while a g++ build
// of the same functions serves the CPU-side unit tests.
//
// Arithmetic is fp32 with two exceptions, both because the reference computes
// in f64 throughout (src/geometry.rs:74) and fp32 would not hold 1e-5 relative
// agreement: sphere intersection (the discriminant of shape.rs:265 cancels for
// the radius-1000 spheres) and primary-ray generation.  Built with
// -ffp-contract=off; fused multiply-adds are written explicitly so that every
// kernel that inlines these functions produces the same bits.
#ifndef RT_CORE_H
#define RT_CORE_H

#include <math.h>
#include <stdint.h>

#include "../../include/rt_abi.h"
#include <cmath>
#include <cstring>
#include <functional>
#include <vector>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RT_HD __host__ __device__ __forceinline__
#else
#define RT_HD inline
#endif
// "Some lane of the wave" on the device, the condition itself on the host: a rare piece of work behind `if (RT_ANY_LANE(c)) if (c)`
// costs the waves that have none of it a scalar branch, where a plain `if (c)` of a few dozen instructions is turned into
// selects that every lane pays for.
#if defined(__HIP_DEVICE_COMPILE__)
#define RT_ANY_LANE(c) (__builtin_amdgcn_ballot_w64(c) != 0ull)
#else
#define RT_ANY_LANE(c) (c)
#endif

// Optional work counters for host-side analysis builds (empty in the product).
#ifndef RT_COUNT
#define RT_COUNT(name)
#endif

namespace rt {

#define RT_PI_F 3.14159265358979323846f
#define RT_PI_D 3.14159265358979323846
#define RT_INF_F (__builtin_inff())
#define RT_NAN_F (__builtin_nanf(""))
#define RT_TAG_CAMERA 0xFFFFFFFFu

struct V3 { float x, y, z; };
RT_HD V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_HD V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_HD V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_HD V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
RT_HD V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
RT_HD V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_HD V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
RT_HD float dot(V3 a, V3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
RT_HD float norm(V3 a) { return sqrtf(dot(a, a)); }
RT_HD V3 normed(V3 a) { return a / norm(a); }  // geometry.rs:102: NaN for the zero vector
RT_HD float comp(V3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
RT_HD V3 madd(V3 o, V3 d, float t) { return v3(fmaf(d.x, t, o.x), fmaf(d.y, t, o.y), fmaf(d.z, t, o.z)); }
// Ray::at (geometry.rs:340-346): t <= 0 returns the origin.
RT_HD V3 ray_at(V3 o, V3 d, float t) { return (t <= 0.0f) ? o : madd(o, d, t); }

struct D3 { double x, y, z; };
RT_HD D3 d3(double x, double y, double z) { D3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_HD D3 d3(V3 a) { return d3((double)a.x, (double)a.y, (double)a.z); }
RT_HD D3 d3(const double* p) { return d3(p[0], p[1], p[2]); }
RT_HD D3 d3(D3 a) { return a; }
RT_HD D3 operator+(D3 a, D3 b) { return d3(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_HD D3 operator-(D3 a, D3 b) { return d3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_HD D3 operator*(D3 a, double s) { return d3(a.x * s, a.y * s, a.z * s); }
RT_HD double dot(D3 a, D3 b) { return fma(a.z, b.z, fma(a.y, b.y, a.x * b.x)); }
RT_HD V3 v3(D3 a) { return v3((float)a.x, (float)a.y, (float)a.z); }

// Rotation about a coordinate axis (geometry.rs:33-50): (c0,c1) = X:(1,2) Y:(2,0) Z:(0,1);
// out[c0] = s[c0] cos - s[c1] sin; out[c1] = s[c0] sin + s[c1] cos.
RT_HD V3 rotate_cs(V3 s, uint32_t kind, float c, float sn) {
    V3 o = s;
    if (kind == RT_XF_ROTATE_X) { o.y = fmaf(s.y, c, -(s.z * sn)); o.z = fmaf(s.y, sn, s.z * c); }
    else if (kind == RT_XF_ROTATE_Y) { o.z = fmaf(s.z, c, -(s.x * sn)); o.x = fmaf(s.z, sn, s.x * c); }
    else { o.x = fmaf(s.x, c, -(s.y * sn)); o.y = fmaf(s.x, sn, s.y * c); }
    return o;
}

// ------------------------------------------------------------------ Philox
struct U4 { uint32_t x, y, z, w; };
RT_HD U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += W0; k1 += W1;
    }
    U4 o; o.x = c0; o.y = c1; o.z = c2; o.w = c3; return o;
}
// Uniform variate in the open interval (0,1): the 2^23 odd multiples of 2^-24, exact in fp32 (and in the
// f64 oracle).  Never 0: the reference draws 53-bit variates, for which z = -1 in its sphere sampler (and the
// zero vector "random_unit + normal" that follows on a z-facing surface, NaN for the whole pixel) has
// probability ~1e-16 per draw; with a 24-bit grid that includes 0 it would be 6e-8, thousands of black
// pixels in a 10 000-spp frame.
RT_HD float u01(uint32_t x) { return (float)((x >> 9) * 2u + 1u) * (1.0f / 16777216.0f); }

// Identity of the path segment being traced; keys every random draw.
struct DrawKey {
    uint32_t k0, k1;  // render seed
    uint32_t pixel, sample, bounce;
    float medium_u;   // >= 0: injected variate used by every medium (parity hook)
};
RT_HD float medium_variate(const DrawKey& k, uint32_t prim) {
    if (k.medium_u >= 0.0f) return k.medium_u;
    return u01(philox4x32_10(k.pixel, k.sample, k.bounce, 1u + prim, k.k0, k.k1).x);
}

// --------------------------------------------------------------- the scene
struct SceneView {
    const rt_prim* prims; const rt_sphere* spheres; const struct DevRect* rects; const struct DevBox* boxes; const rt_xform* xforms;  // rects, boxes: make_device_shapes
    const struct ShadeRec* shade;  // per primitive: the primitive and its material side by side (make_shade_records)
    const rt_texture* textures; const rt_image* images; const uint8_t* texels;
    const uint8_t* perm; const struct WalkNode* bvh;  // bvh: n_bvh_nodes slots and one sentinel slot (make_walk_nodes)
    uint32_t n_prims, n_bvh_nodes;
    float background[3];
    // array lengths, for the kernels that copy a small scene's tables into LDS (texels stay in global memory)
    uint32_t n_spheres, n_rects, n_boxes, n_xforms, n_textures, n_images, n_perm_tables;
    uint32_t lds_scene_bytes;  // bytes those tables take in LDS behind the node array (each 16-byte aligned); 0 = leave them in global memory
    uint32_t side_rounding;    // some primitive's hit points need side_of_surface (scene_needs_side_rounding); the kernels come in two builds, with and without
    uint32_t fp32_planes;      // no medium can be hit in this scene: the hit points of unwrapped boxes and rectangles are formed in fp32 (prim_surface)
    uint32_t multi_scatter;    // some medium is flagged RT_PRIM_MULTI_SCATTER; the traversal kernel has a build without that code (the flag is then ignored: the host picks the build)
};

// Rectangles and boxes as the kernels hold them: every f64 coordinate of the description as an fp32 value `hi` and an
// fp32 remainder `lo` (hi + lo carries 48 bits of it).  A ray that starts close to a plane gets its t from the small
// difference (plane - origin); formed as (hi - origin) + lo that difference keeps its relative accuracy in fp32, where
// a plane rounded to fp32 alone would cost up to 1e-5 of t.  The f64 code paths use hi + lo as the coordinate.
struct DevRect { uint32_t axis; float x0, y0, x1, y1, h; float x0_lo, y0_lo, x1_lo, y1_lo, h_lo; uint32_t pad; };
struct DevBox { float min[3], max[3], min_lo[3], max_lo[3]; };
static_assert(sizeof(DevRect) == 48 && sizeof(DevBox) == 48, "device shape sizes");
inline void split_hi_lo(double x, float& hi, float& lo) {
    hi = (float)x;
    lo = (hi - hi == 0.0f) ? (float)(x - (double)hi) : 0.0f;  // no remainder for infinities and NaN
}
// (host side) the device forms of a description's rectangles and boxes
inline void make_device_shapes(const rt_scene_desc* D, std::vector<DevRect>& rects, std::vector<DevBox>& boxes) {
    rects.resize(D->n_rects); boxes.resize(D->n_boxes);
    for (uint32_t i = 0; i < D->n_rects; ++i) {
        const rt_rect& R = D->rects[i]; DevRect& o = rects[i];
        o.axis = R.axis; o.pad = 0;
        split_hi_lo(R.x0, o.x0, o.x0_lo); split_hi_lo(R.y0, o.y0, o.y0_lo); split_hi_lo(R.x1, o.x1, o.x1_lo); split_hi_lo(R.y1, o.y1, o.y1_lo);
        split_hi_lo(R.height, o.h, o.h_lo);
    }
    for (uint32_t i = 0; i < D->n_boxes; ++i)
        for (int c = 0; c < 3; ++c) { split_hi_lo(D->boxes[i].min[c], boxes[i].min[c], boxes[i].min_lo[c]); split_hi_lo(D->boxes[i].max[c], boxes[i].max[c], boxes[i].max_lo[c]); }
}

RT_HD void axis_perm(uint32_t axis, int& p1, int& p2, int& o) {  // shape.rs:138-142
    if (axis == 0) { p1 = 1; p2 = 2; o = 0; } else if (axis == 1) { p1 = 2; p2 = 0; o = 1; } else { p1 = 0; p2 = 1; o = 2; }
}

// Geometry below is written once for two precisions.  A primitive without Translated / Rotated
// wrappers is tested in fp32: (height - origin) is then formed from the ray's own coordinates and is
// accurate to an ulp of the difference.  Behind a wrapper the origin is first moved into the
// shape's frame, which in fp32 would cost an ulp of the COORDINATE (1e-5 at 300 units, unbounded in t
// for grazing rays), so wrapped primitives take the same code in f64 (sin/cos and offsets are kept in
// f64 in rt_xform), as spheres always do.
RT_HD double compd(D3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
// (f64: the product and the sum rounded one after the other, as the reference's `origin + direction * t` is, geometry.rs:345.
// The sign of the few-ulp residue a hit point keeps off its plane is observable: a ray leaving the floor under a medium
// whose boundary box stands on it starts inside the box for a residue >= 0 and outside for < 0, and only then does the
// medium act on it (shape.rs:386-410).  Unfused, the residue is exactly 0 more often than not, as in the reference; a
// fused multiply-add leaves it non-zero, half of the time negative, and darkens cornell_smoke's floor by up to 14 %.)
RT_HD D3 ray_at(D3 o, D3 d, double t) { return (t <= 0.0) ? o : d3(d.x * t + o.x, d.y * t + o.y, d.z * t + o.z); }
// Ray::at with the clamp at t <= 0 left out on request (the boundary queries of a medium flagged RT_PRIM_MULTI_SCATTER, which is
// not the reference's behaviour).
RT_HD V3 ray_at_opt(V3 o, V3 d, float t, bool unclamped) { return unclamped ? madd(o, d, t) : ray_at(o, d, t); }
RT_HD D3 ray_at_opt(D3 o, D3 d, double t, bool unclamped) { return unclamped ? d3(d.x * t + o.x, d.y * t + o.y, d.z * t + o.z) : ray_at(o, d, t); }
RT_HD D3 rotate_cs(D3 s, uint32_t kind, double c, double sn) {
    D3 o = s;
    if (kind == RT_XF_ROTATE_X) { o.y = s.y * c - s.z * sn; o.z = s.y * sn + s.z * c; }
    else if (kind == RT_XF_ROTATE_Y) { o.z = s.z * c - s.x * sn; o.x = s.z * sn + s.x * c; }
    else { o.x = s.x * c - s.y * sn; o.y = s.x * sn + s.y * c; }
    return o;
}
RT_HD float get(V3 a, int i) { return comp(a, i); }
RT_HD double get(D3 a, int i) { return compd(a, i); }
RT_HD void put(V3& a, int i, float v) { if (i == 0) a.x = v; else if (i == 1) a.y = v; else a.z = v; }
RT_HD void put(D3& a, int i, double v) { if (i == 0) a.x = v; else if (i == 1) a.y = v; else a.z = v; }
RT_HD float norm_of(V3 a) { return norm(a); }
RT_HD double norm_of(D3 a) { return sqrt(dot(a, a)); }
RT_HD float log_of(float x) { return logf(x); }
RT_HD double log_of(double x) { return (double)logf((float)x); }
template <class T> struct Lim;
template <> struct Lim<float> { RT_HD static float inf() { return RT_INF_F; } RT_HD static float mx(float a, float b) { return fmaxf(a, b); } RT_HD static float mn(float a, float b) { return fminf(a, b); } };
template <> struct Lim<double> { RT_HD static double inf() { return (double)RT_INF_F; } RT_HD static double mx(double a, double b) { return fmax(a, b); } RT_HD static double mn(double a, double b) { return fmin(a, b); } };

// Ray into the frame of the innermost shape: Translated subtracts the offset from the origin
// (shape.rs:356-360), Rotated rotates origin and direction by -angle (shape.rs:370-374); wrappers are
// applied outermost first.
RT_HD void to_local(const SceneView& S, const rt_prim& P, D3& o, D3& d) {
    for (uint32_t k = 0; k < P.xf_count; ++k) {
        const rt_xform X = S.xforms[P.xf_first + k];
        if (X.kind == RT_XF_TRANSLATE) o = o - d3(X.a, X.b, X.c);
        else { o = rotate_cs(o, X.kind, X.a, -X.b); d = rotate_cs(d, X.kind, X.a, -X.b); }
    }
}

// Sphere / MovingSphere, near root only (shape.rs:256-307), in f64.
RT_HD D3 sphere_center(const rt_sphere& sp, bool moving, float time) {
    D3 c = d3(sp.center);
    if (moving) {
        double dt = (double)time - sp.start_time;  // MovingSphere: ray.at(time - start_time), clamped at <= 0
        if (!(dt <= 0.0)) c = c + d3(sp.motion) * dt;
    }
    return c;
}
// Square root and reciprocal in f64 for the sphere's root.  On the GPU: the hardware estimate (good to 2^-23) refined to
// within an ulp or two for the square root and to 2^-46 for the reciprocal, a fifth of the instructions of the correctly
// rounded library routines; the root is rounded to fp32 afterwards, 2^22 times coarser.  On the host: the plain operations.
#if defined(__HIP_DEVICE_COMPILE__)
RT_HD double sqrt_near(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;  // (one coupled step squares the estimate's 2^-23, the correction squares that again)
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    g = fma(fma(-g, g, x), h, g);
    return x == 0.0 ? 0.0 : g;
}
RT_HD double rcp_near(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);  // (one step: 2^-46 from an estimate good to 2^-23)
    return r;
}
RT_HD double rsqrt_near(double x) {  // 1 / sqrt(x) for positive finite x
    const double y = __builtin_amdgcn_rsq(x);
    double r = fma(-x * y, y, 1.0);           // 1 - x y^2
    double z = fma(y * r, fma(r, 0.375, 0.5), y);
    r = fma(-x * z, z, 1.0);
    return fma(z * r, 0.5, z);
}
// 1 / x in f64 for any x: the refinement of rcp_near turns an infinite estimate into NaN (0 * inf), so zeros and
// infinities keep the hardware's own answer (+-inf, +-0), which is the quotient's.
RT_HD double rcp_exact_at_zero(double x) {
    const double r0 = __builtin_amdgcn_rcp(x);
    double r = fma(fma(-x, r0, 1.0), r0, r0);
    r = fma(fma(-x, r, 1.0), r, r);
    return (r - r == 0.0) ? r : r0;  // (finite: the refined value; else the estimate)
}
// 1 / direction for the walk (the hardware reciprocal, within an ulp; the walk's node boxes are widened by far more, and the
// exact leaf tests of one ray all use the same value)
RT_HD V3 ray_inv(V3 d) { return v3(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z)); }
// sine and cosine of 2 pi u, 0 < u < 1 (the argument is reduced exactly in turns, with no large-argument path)
RT_HD void sincos_turns(float u, float& s, float& c) { sincospif(2.0f * u, &s, &c); }
#else
RT_HD double sqrt_near(double x) { return sqrt(x); }
RT_HD double rcp_near(double x) { return 1.0 / x; }
RT_HD double rsqrt_near(double x) { return 1.0 / sqrt(x); }
RT_HD double rcp_exact_at_zero(double x) { return 1.0 / x; }
RT_HD V3 ray_inv(V3 d) { return v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z); }
RT_HD void sincos_turns(float u, float& s, float& c) { const float phi = (2.0f * RT_PI_F) * u; s = sinf(phi); c = cosf(phi); }
#endif
// The walk's node test forms (bound * inv - origin * inv); with an infinite inv (a zero direction component) that is
// inf - inf = NaN for every bound on the origin's side of zero, and the test would take the slab for unbounded where the
// reference (shape.rs:584-588: (bound - origin) / 0 = +-inf) sees the ray pass beside the box.  The walk therefore caps
// |inv| at 2^100: the products stay finite for coordinates up to 1e8 and keep the signs of (bound - origin), and any
// finite capped quotient is beyond every hit.  NaN stays NaN (a NaN direction component makes the reference ignore the
// axis).  The leaf tests get the infinity back (walk_uncap): they follow the reference's quotients literally.
#define RT_INV_CAP 1.2676506002282294e30f
// A capped quotient is finite where the reference has +inf; a slab whose near end is there is missed by the reference
// (tmax <= tmin with both infinite, shape.rs:593), so the node tests run against min(tmax, 2^90): no hit lies beyond.
#define RT_WALK_FAR 1.2379400392853803e27f
RT_HD float cap_inv(float r) { return fabsf(r) > RT_INV_CAP ? copysignf(RT_INV_CAP, r) : r; }
// (two exact scalings instead of a comparison: 2^100 * 2^28 overflows to infinity, everything below comes back unchanged)
RT_HD float uncap_inv(float c) { return (c * 268435456.0f) * 3.7252902984619140625e-9f; }
// A direction component of -0.0: the reference swaps a slab's two quotients only for d < 0.0 (shape.rs:586), which -0.0
// is not, so the quotients (bound - origin) / -0.0 = -+inf stay in the order (far, near) and every bounding box with an
// extent on that axis is missed: such a ray hits nothing that stands behind a bounding-box test (everything in a scene).
RT_HD bool has_negative_zero(V3 d) {
    uint32_t x, y, z; memcpy(&x, &d.x, 4); memcpy(&y, &d.y, 4); memcpy(&z, &d.z, 4);
    return x == 0x80000000u || y == 0x80000000u || z == 0x80000000u;
}
RT_HD V3 walk_cap(V3 inv) { return v3(cap_inv(inv.x), cap_inv(inv.y), cap_inv(inv.z)); }
RT_HD V3 walk_uncap(V3 inv) { return v3(uncap_inv(inv.x), uncap_inv(inv.y), uncap_inv(inv.z)); }
// Near root (-half_b - sqrt(disc)) / a of shape.rs:262-270; false when the discriminant is negative.
RT_HD bool sphere_root(const rt_sphere& sp, bool moving, D3 o, D3 dd, float time, double& t_out) {
    D3 c = sphere_center(sp, moving, time);
    D3 oc = o - c;
    double a = dot(dd, dd);
    double half_b = dot(oc, dd);
    double cc = dot(oc, oc) - sp.radius * sp.radius;
    double disc = half_b * half_b - a * cc;
    if (disc < 0.0) return false;
    t_out = (-half_b - sqrt_near(disc)) * rcp_near(a);
    return true;
}

// The far root (-half_b + sqrt(disc)) / a, for spheres flagged RT_PRIM_BOTH_ROOTS (not the reference's behaviour); the
// discriminant is known not to be negative.
RT_HD double sphere_far_root(const rt_sphere& sp, bool moving, D3 o, D3 dd, float time) {
    D3 oc = o - sphere_center(sp, moving, time);
    double a = dot(dd, dd), half_b = dot(oc, dd), cc = dot(oc, oc) - sp.radius * sp.radius;
    return (-half_b + sqrt_near(half_b * half_b - a * cc)) * rcp_near(a);
}

// Rectangle (shape.rs:309-332): inclusive interval and extents tests written so that NaN and
// infinity flow exactly as in the reference.
// `height_lo`: the remainder of the plane's coordinate (fp32 callers; the f64 callers pass the whole coordinate as `height`).
template <class T, class VT, bool UNCLAMPED = false>
RT_HD bool rect_t(uint32_t axis, T x0, T y0, T x1, T y1, T height, T height_lo, VT o, VT d, T tmin, T tmax, T& t_out) {
    int p1, p2, oo; axis_perm(axis, p1, p2, oo);
    T t = ((height - get(o, oo)) + height_lo) / get(d, oo);
    if (t < tmin || t > tmax) return false;
    VT pt = ray_at_opt(o, d, t, UNCLAMPED);
    T a = get(pt, p1), b = get(pt, p2);
    if (a < x0 || a > x1 || b < y0 || b > y1) return false;
    t_out = t;
    return true;
}
// Side k of Shape::new_box (shape.rs:157-215): x0(F) x1 y0(F) y1 z0(F) z1.  A Rectangle reads its
// (lower_left, upper_right) pairs in (p1, p2) order, which is (z, x) for Axis::Y (shape.rs:315-319), but
// new_box fills the y sides with (x0, z0) / (x1, z1) (shape.rs:180-195): on those two sides the hit
// point's z is tested against the box's x range and its x against the z range.  Reproduced as is.
template <class T> RT_HD T coord(float hi, float lo);
template <> RT_HD float coord<float>(float hi, float) { return hi; }
template <> RT_HD double coord<double>(float hi, float lo) { return (double)hi + (double)lo; }
// (components picked with selects: indexing the arrays with a run-time side would put the box in scratch memory)
template <class T>
RT_HD void box_side(const DevBox& B, int k, uint32_t& axis, T& x0, T& y0, T& x1, T& y1, T& h, T& h_lo) {
    axis = (uint32_t)(k >> 1);
    int p1, p2, oo; axis_perm(axis, p1, p2, oo);
    const V3 mn = v3(B.min[0], B.min[1], B.min[2]), mx = v3(B.max[0], B.max[1], B.max[2]);
    const V3 mnl = v3(B.min_lo[0], B.min_lo[1], B.min_lo[2]), mxl = v3(B.max_lo[0], B.max_lo[1], B.max_lo[2]);
    if (axis == 1) p1 = 0, p2 = 2;  // the swapped extents of the y sides
    x0 = coord<T>(comp(mn, p1), comp(mnl, p1)); y0 = coord<T>(comp(mn, p2), comp(mnl, p2));
    x1 = coord<T>(comp(mx, p1), comp(mxl, p1)); y1 = coord<T>(comp(mx, p2), comp(mxl, p2));
    const float hh = (k & 1) ? comp(mx, oo) : comp(mn, oo), hl = (k & 1) ? comp(mxl, oo) : comp(mnl, oo);
    h = coord<T>(hh, hl);
    h_lo = sizeof(T) == sizeof(float) ? (T)hl : (T)0;  // already inside h for f64
}
// Box = Vec<Shape>::intersect over the six sides with shrinking tmax (shape.rs:17-34):
// the smallest accepted t wins and a later side wins an exact tie.
template <class T, class VT, bool UNCLAMPED = false>
RT_HD bool box_t(const DevBox& B, VT o, VT d, T tmin, T tmax, T& t_out, int& side) {
    bool any = false;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        uint32_t axis; T x0, y0, x1, y1, h, h_lo, t;
        box_side<T>(B, k, axis, x0, y0, x1, y1, h, h_lo);
        if (rect_t<T, VT, UNCLAMPED>(axis, x0, y0, x1, y1, h, h_lo, o, d, tmin, tmax, t)) { tmax = t; t_out = t; side = k; any = true; }
    }
    return any;
}
// The same six sides in fp32 with the plane parameters given: q[2a + s] = ((bound[s][a] - origin[a]) + remainder) * (1 / direction[a]).
// Every fp32 box test and every exact leaf test (closest_hit) forms its parameters by this one expression, so two
// boxes that share a face see bit-identical t on it, which is what the reference's tie rule needs.
RT_HD void box_quotients(const DevBox& B, V3 o, V3 inv, float* q) {
    q[0] = ((B.min[0] - o.x) + B.min_lo[0]) * inv.x; q[1] = ((B.max[0] - o.x) + B.max_lo[0]) * inv.x;
    q[2] = ((B.min[1] - o.y) + B.min_lo[1]) * inv.y; q[3] = ((B.max[1] - o.y) + B.max_lo[1]) * inv.y;
    q[4] = ((B.min[2] - o.z) + B.min_lo[2]) * inv.z; q[5] = ((B.max[2] - o.z) + B.max_lo[2]) * inv.z;
}
template <bool UNCLAMPED = false>
RT_HD bool box_sides_q(const DevBox& B, V3 o, V3 d, const float* q, float tmin, float tmax, float& t_out, int& side) {
    bool any = false;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        uint32_t axis; float x0, y0, x1, y1, h, h_lo;
        box_side<float>(B, k, axis, x0, y0, x1, y1, h, h_lo);
        int p1, p2, oo; axis_perm(axis, p1, p2, oo);
        const float t = q[k];
        const V3 pt = ray_at_opt(o, d, t, UNCLAMPED);
        const float a = comp(pt, p1), b = comp(pt, p2);
        // no branches: the six sides run as straight code, their conditions as flags
        const bool ok = !(t < tmin || t > tmax) && !(a < x0 || a > x1 || b < y0 || b > y1);
        tmax = ok ? t : tmax; t_out = ok ? t : t_out; side = ok ? k : side; any = any || ok;
    }
    return any;
}
template <bool UNCLAMPED>
RT_HD bool box_f32(const DevBox& B, V3 o, V3 d, float tmin, float tmax, float& t_out, int& side) {
    const V3 inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    float q[6];
    box_quotients(B, o, inv, q);
    return box_sides_q<UNCLAMPED>(B, o, d, q, tmin, tmax, t_out, side);
}
template <> RT_HD bool box_t<float, V3, false>(const DevBox& B, V3 o, V3 d, float tmin, float tmax, float& t_out, int& side) { return box_f32<false>(B, o, d, tmin, tmax, t_out, side); }
template <> RT_HD bool box_t<float, V3, true>(const DevBox& B, V3 o, V3 d, float tmin, float tmax, float& t_out, int& side) { return box_f32<true>(B, o, d, tmin, tmax, t_out, side); }

// The f64 form (boxes behind Translated / Rotated wrappers, boundaries of media): the same six sides with the plane parameters
// formed as (bound - origin) * (1 / direction), the three reciprocals refined to within an ulp or two (rcp_exact_at_zero).  The
// reference divides (shape.rs:321); the product differs from the quotient by a couple of f64 ulps, nine orders of magnitude
// below the fp32 rounding of the accepted t, and goes to +-inf / NaN for a zero direction component exactly as the quotient
// does.  Six f64 divisions were two thirds of the instructions of this test.
template <bool UNCLAMPED>
RT_HD bool box_f64(const DevBox& B, D3 o, D3 d, double tmin, double tmax, double& t_out, int& side) {
    const D3 inv = d3(rcp_exact_at_zero(d.x), rcp_exact_at_zero(d.y), rcp_exact_at_zero(d.z));
    bool any = false;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        uint32_t axis; double x0, y0, x1, y1, h, h_lo;
        box_side<double>(B, k, axis, x0, y0, x1, y1, h, h_lo);
        int p1, p2, oo; axis_perm(axis, p1, p2, oo);
        const double t = (h - compd(o, oo)) * compd(inv, oo);
        const D3 pt = ray_at_opt(o, d, t, UNCLAMPED);
        const double a = compd(pt, p1), b = compd(pt, p2);
        const bool ok = !(t < tmin || t > tmax) && !(a < x0 || a > x1 || b < y0 || b > y1);
        tmax = ok ? t : tmax; t_out = ok ? t : t_out; side = ok ? k : side; any = any || ok;
    }
    return any;
}
template <> RT_HD bool box_t<double, D3, false>(const DevBox& B, D3 o, D3 d, double tmin, double tmax, double& t_out, int& side) { return box_f64<false>(B, o, d, tmin, tmax, t_out, side); }
template <> RT_HD bool box_t<double, D3, true>(const DevBox& B, D3 o, D3 d, double tmin, double tmax, double& t_out, int& side) { return box_f64<true>(B, o, d, tmin, tmax, t_out, side); }

// Ray parameter of the innermost shape for a ray already in its frame.
template <class T, class VT, bool UNCLAMPED = false>
RT_HD bool geom_t(const SceneView& S, const rt_prim& P, VT o, VT d, T tmin, T tmax, float time, T& t, int& side) {
    side = 0;
    if (P.geom == RT_GEOM_SPHERE || P.geom == RT_GEOM_MOVING_SPHERE) {
        double t64;
        if (!sphere_root(S.spheres[P.geom_index], P.geom == RT_GEOM_MOVING_SPHERE, d3(o), d3(d), time, t64)) return false;
        if (!(t64 > (double)tmin && t64 < (double)tmax)) {  // strict on both ends (shape.rs:271)
            if (!(P.flags & RT_PRIM_BOTH_ROOTS)) return false;
            t64 = sphere_far_root(S.spheres[P.geom_index], P.geom == RT_GEOM_MOVING_SPHERE, d3(o), d3(d), time);  // (the variant: then the far root)
            if (!(t64 > (double)tmin && t64 < (double)tmax)) return false;
        }
        t = (T)t64;
        return true;
    }
    if (P.geom == RT_GEOM_RECT) {
        const DevRect R = S.rects[P.geom_index];
        return rect_t<T, VT, UNCLAMPED>(R.axis, coord<T>(R.x0, R.x0_lo), coord<T>(R.y0, R.y0_lo), coord<T>(R.x1, R.x1_lo), coord<T>(R.y1, R.y1_lo), coord<T>(R.h, R.h_lo),
                                        sizeof(T) == sizeof(float) ? (T)R.h_lo : (T)0, o, d, tmin, tmax, t);
    }
    return box_t<T, VT, UNCLAMPED>(S.boxes[P.geom_index], o, d, tmin, tmax, t, side);
}

// Intersection::intersect reduced to the accepted ray parameter (plus the box side) for a ray in the
// shape's frame.  ConstantMedium follows shape.rs:386-410 step by step.
template <class T, class VT>
RT_HD bool local_t(const SceneView& S, const rt_prim& P, uint32_t prim, VT o, VT d, T tmin, T tmax, float time, const DrawKey& key, T& t, int& side) {
    if (!(P.flags & RT_PRIM_MEDIUM)) return geom_t<T, VT>(S, P, o, d, tmin, tmax, time, t, side);
    T t1, t2; int s_;
    // (RT_PRIM_MULTI_SCATTER, not the reference's behaviour: the two boundary queries see the points behind the origin where they
    // are, so a ray that starts inside the boundary finds the way in behind it and the way out ahead of it, and scatters again)
    if (S.multi_scatter != 0u && (P.flags & RT_PRIM_MULTI_SCATTER)) {
        if (!geom_t<T, VT, true>(S, P, o, d, -Lim<T>::inf(), Lim<T>::inf(), time, t1, s_)) return false;
        if (!geom_t<T, VT, true>(S, P, o, d, t1 + (T)0.001, Lim<T>::inf(), time, t2, s_)) return false;
    } else {
        if (!geom_t<T, VT>(S, P, o, d, -Lim<T>::inf(), Lim<T>::inf(), time, t1, s_)) return false;
        if (!geom_t<T, VT>(S, P, o, d, t1 + (T)0.001, Lim<T>::inf(), time, t2, s_)) return false;
    }
    t1 = Lim<T>::mx(t1, tmin);
    t2 = Lim<T>::mn(t2, tmax);
    if (t1 >= t2) return false;
    t1 = Lim<T>::mx(t1, (T)0);
    T length = norm_of(d);
    T inside = (t2 - t1) * length;
    T hit_distance = (T)-1 / (T)P.density * log_of((T)medium_variate(key, prim));
    if (hit_distance > inside) return false;
    t = t1 + hit_distance / length;
    side = 0;
    return true;
}
RT_HD bool prim_t(const SceneView& S, uint32_t prim, V3 o, V3 d, float tmin, float tmax, float time, const DrawKey& key, float& t, int& side) {
    const rt_prim P = S.prims[prim];
    if (P.xf_count == 0) return local_t<float, V3>(S, P, prim, o, d, tmin, tmax, time, key, t, side);
    D3 lo = d3(o), ld = d3(d);
    to_local(S, P, lo, ld);
    double t64;
    if (!local_t<double, D3>(S, P, prim, lo, ld, (double)tmin, (double)tmax, time, key, t64, side)) return false;
    t = (float)t64;
    return true;
}

struct Hit {  // IntersectionPoint (shape.rs:539-550)
    float t, u, v;
    V3 point, normal, in_vec;
    uint32_t prim;
    uint32_t point_side = 0;  // where the f64 point lies from `point`, for the surfaces where shade_step needs it (see side_of_surface): bit 2c = above on axis c, bit 2c + 1 = below; zero elsewhere
};

// Point, normal and surface coordinates in the shape's frame for an accepted (t, side).  Runs once per
// path segment (not per candidate), in f64: the ray parameter is re-derived in f64 (t is its fp32
// rounding), so the point is the reference's point rounded once rather than origin + fp32(t) * direction.
RT_HD void local_surface(const SceneView& S, const rt_prim& P, D3 o, D3 d, float time, float t, int side, bool want_uv, D3& pt_out, D3& n_out, double& u, double& v) {
    // (every branch writes the same scalar locals and the results are put together once at the end: stores to
    // different fields of the caller's structs in different branches end up in scratch memory)
    double px, py, pz, nx, ny, nz, uu = 0.0, vv = 0.0;
    if (P.flags & RT_PRIM_MEDIUM) {
        const D3 p = ray_at(o, d, (double)t);
        px = p.x; py = p.y; pz = p.z;
        nx = 1.0; ny = 0.0; nz = 0.0;  // shape.rs:413
    } else if (P.geom == RT_GEOM_SPHERE || P.geom == RT_GEOM_MOVING_SPHERE) {
        const rt_sphere sp = S.spheres[P.geom_index];
        const bool moving = P.geom == RT_GEOM_MOVING_SPHERE;
        // The accepted t is the root rounded to fp32: taken as it is, the point lies up to |d| t 2^-24 along the ray from
        // the reference's, and the normal is off by that over the radius.  Where that could reach 1e-6 (t |d| > 16 r: small
        // spheres seen from afar), and where surface coordinates are wanted (near a pole v moves by the square root of what
        // the point moves by), the root is derived again in f64; elsewhere -- the wave as a whole, most of the time -- the
        // rounded t does (the next ray's origin is rounded to fp32 anyway, which moves it further than this).
        double t64 = (double)t;
        const bool again = want_uv || t64 * t64 * dot(d, d) > 256.0 * sp.radius * sp.radius || !(t64 - t64 == 0.0);
        if (RT_ANY_LANE(again)) if (again) {
            sphere_root(sp, moving, o, d, time, t64);
            if (P.flags & RT_PRIM_BOTH_ROOTS) {  // the accepted t is the fp32 rounding of one of the two roots
                const double far_root = sphere_far_root(sp, moving, o, d, time);
                if (fabs(far_root - (double)t) < fabs(t64 - (double)t)) t64 = far_root;
            }
        }
        const D3 c = sphere_center(sp, moving, time);
        const D3 p = ray_at(o, d, t64);
        const D3 pc = p - c;
        const double inv_len = rsqrt_near(dot(pc, pc));
        nx = pc.x * inv_len; ny = pc.y * inv_len; nz = pc.z * inv_len;
        if (want_uv) {
            const D3 q = pc * (1.0 / sp.radius);  // shape.rs:275 divides by the signed radius
            const float phi = atan2f((float)q.z, (float)q.x);
            // theta = asin(q.y) as atan2(q.y, sqrt(1 - q.y^2)), the square root in f64: exact for every |q.y| <= 1,
            // NaN beyond like asin, and without fp32 asin's loss of accuracy near the poles
            const float theta = atan2f((float)q.y, (float)sqrt(1.0 - q.y * q.y));
            uu = (double)(1.0f - (phi + RT_PI_F) / (2.0f * RT_PI_F));
            vv = (double)((theta + RT_PI_F / 2.0f) / RT_PI_F);
        }
        px = p.x; py = p.y; pz = p.z;
    } else {
        // the plane first; the extents are only looked at for (u, v)
        uint32_t axis; double hgt; bool flip_side = false;
        if (P.geom == RT_GEOM_RECT) {
            const DevRect& R = S.rects[P.geom_index];
            axis = R.axis; hgt = coord<double>(R.h, R.h_lo);
        } else {
            const DevBox& B = S.boxes[P.geom_index];
            axis = (uint32_t)(side >> 1);
            const bool upper = (side & 1) != 0;
            const float hh = axis == 0 ? (upper ? B.max[0] : B.min[0]) : (axis == 1 ? (upper ? B.max[1] : B.min[1]) : (upper ? B.max[2] : B.min[2]));
            const float hl = axis == 0 ? (upper ? B.max_lo[0] : B.min_lo[0]) : (axis == 1 ? (upper ? B.max_lo[1] : B.min_lo[1]) : (upper ? B.max_lo[2] : B.min_lo[2]));
            hgt = coord<double>(hh, hl);
            flip_side = !upper;  // the x0 / y0 / z0 sides are Flipped (outward normals)
        }
        int p1, p2, oo; axis_perm(axis, p1, p2, oo);
        // shape.rs:321.  A true division: a ray parallel to the plane must give +-inf or NaN as there, and the point's coordinate
        // on the plane's axis, origin + quotient * direction, must cancel to exactly the plane's as often as the reference's does
        // (a product with a refined reciprocal is two ulps off and cancels half as often, which shifts cornell_smoke's frame
        // mean by two of its standard errors: the smoke boxes stand on the floor's plane)
        const double t64 = (hgt - compd(o, oo)) / compd(d, oo);
        const D3 p = ray_at(o, d, t64);
        if (want_uv) {
            double x0, y0, x1, y1, h_, unused;
            if (P.geom == RT_GEOM_RECT) {
                const DevRect& R = S.rects[P.geom_index];
                x0 = coord<double>(R.x0, R.x0_lo); y0 = coord<double>(R.y0, R.y0_lo); x1 = coord<double>(R.x1, R.x1_lo); y1 = coord<double>(R.y1, R.y1_lo);
            } else {
                uint32_t axis_;
                box_side<double>(S.boxes[P.geom_index], side, axis_, x0, y0, x1, y1, h_, unused);
            }
            uu = (compd(p, p1) - x0) / (x1 - x0);
            vv = (compd(p, p2) - y0) / (y1 - y0);
        }
        const double s = flip_side ? -1.0 : 1.0;
        px = p.x; py = p.y; pz = p.z;
        nx = oo == 0 ? s : 0.0; ny = oo == 1 ? s : 0.0; nz = oo == 2 ? s : 0.0;
    }
    pt_out = d3(px, py, pz); n_out = d3(nx, ny, nz); u = uu; v = vv;
}

// A ray's origin is kept in fp32.  The reference's f64 hit point lies on its plane to an ulp of f64, so a ray that leaves
// a surface never meets that surface again: the quotient (plane - origin) / direction is far below tmin = 0.001.  Rounded to
// fp32 the point of a plane that is not aligned with the axes (a box or rectangle behind Rotated / Translated) lies up to
// 2e-5 off it, half of the time behind it, and directions are not normalised (normal + unit vector: a length of 0.05 is
// common), so 0.001 in the ray's parameter can be less than that: 0.3 % of the rays leaving Cornell's tall box met the face
// they started on, mostly rays along the face, among them the ones that go straight up to the light: the box's sides came
// out 3 % darker and the frame 0.34 % (tools/bias_check.py; rounding the origins of the f64 oracle to fp32 shows the same).
// For those surfaces the origin is therefore rounded to the side of the plane the new ray leaves on: a component whose
// fp32 value fell short of the f64 point in that direction moves one fp32 step along it.  Axis-aligned planes at fp32
// coordinates (every wall, the sides of final_scene's boxes) are exact and untouched; the tops of those boxes stand at random
// f64 heights below 101, half an fp32 ulp there is 4e-6, and the oracle with fp32 origins renders final_scene's frame mean
// to within 1e-6 of itself: plain boxes are left alone.  Spheres never meet a ray from within (near root only), and a
// medium has no surface.
// (The same holds for a sphere that is also hit from within, RT_PRIM_BOTH_ROOTS, which is not the reference's behaviour: its
// far root is the surface a ray has just left.)
RT_HD bool plane_off_grid(const rt_prim& P) {
    if (P.flags & RT_PRIM_MEDIUM) return false;
    if (P.geom == RT_GEOM_BOX || P.geom == RT_GEOM_RECT) return P.xf_count > 0;
    return (P.flags & RT_PRIM_BOTH_ROOTS) != 0;
}
inline bool scene_needs_side_rounding(const rt_scene_desc* D) {  // (host side)
    for (uint32_t i = 0; i < D->n_prims; ++i) if (plane_off_grid(D->prims[i])) return true;
    return false;
}
inline bool scene_has_multi_scatter_media(const rt_scene_desc* D) {  // (host side)
    for (uint32_t i = 0; i < D->n_prims; ++i) if ((D->prims[i].flags & RT_PRIM_MEDIUM) && (D->prims[i].flags & RT_PRIM_MULTI_SCATTER)) return true;
    return false;
}
inline bool scene_tree_holds_a_medium(const rt_scene_desc* D) {  // (host side) some leaf's primitive is a ConstantMedium (inert media are left out of the tree)
    for (uint32_t i = 0; i < D->n_bvh_nodes; ++i)
        if (D->bvh[i].prim != RT_BVH_NONE && (D->prims[D->bvh[i].prim & ~RT_BVH_EXACT].flags & RT_PRIM_MEDIUM)) return true;
    return false;
}
RT_HD float fp32_step(float f, bool up) {  // the neighbour of a finite f above / below it
    uint32_t b; memcpy(&b, &f, sizeof(b));
    if ((b & 0x7fffffffu) == 0u) b = up ? 1u : 0x80000001u;
    else b += ((b >> 31) == 0u) == up ? 1u : 0xffffffffu;
    memcpy(&f, &b, sizeof(f));
    return f;
}
// `point` = the f64 point rounded to nearest, `side` = where the f64 point lies from it (Hit::point_side; a NaN sets no bit),
// `n` the surface normal, `leaves_along_n`: the new ray leaves on the normal's side.
RT_HD V3 side_of_surface(V3 point, uint32_t side, V3 n, bool leaves_along_n) {
    float p[3] = {point.x, point.y, point.z};
    const float m[3] = {n.x, n.y, n.z};
    for (int c = 0; c < 3; ++c) {
        const bool up = (m[c] > 0.0f) == leaves_along_n;                       // the direction of "outside" on this axis
        if (m[c] != 0.0f && ((side >> (2 * c + (up ? 0 : 1))) & 1u)) p[c] = fp32_step(p[c], up);
    }
    return v3(p[0], p[1], p[2]);
}

// The rest of the IntersectionPoint for an accepted (prim, t, side), carried back out through the
// wrappers (Rotated rotates point / normal / in_vec by +angle, shape.rs:378-383; Translated adds the
// offset to the point, shape.rs:364-367).
// want_uv / want_in_vec = false leave (u, v) = (0, 0) / in_vec unset for callers that know the material reads neither.
RT_HD void prim_surface(const SceneView& S, const rt_prim& P, uint32_t prim, V3 o, V3 d, float time, float t, int side, bool want_uv, bool want_in_vec, Hit& h) {
    h.t = t; h.prim = prim;
    h.in_vec = want_in_vec ? normed(d) : d;
    // An unwrapped box or rectangle whose material reads no surface coordinates, in a scene without media: the point in fp32.
    // The reference's point lies on the plane to an ulp or two of f64 (origin + direction * t, t the quotient of shape.rs:321), so
    // its coordinate across the plane, rounded to fp32, is the plane's (for a plane that is not an fp32 number: the fp32 number
    // next to it); the other two are origin + direction * t with the t the walk accepted (within two fp32 ulps of the quotient).
    // What is lost is the residue of a few f64 ulps that the reference's point keeps off a plane standing at 0 (fp32 holds
    // numbers that small).  Only a ConstantMedium can tell: its boundary queries run over (-inf, inf) and take a rectangle
    // behind the ray for hit when the ray's ORIGIN lies within the extents (shape.rs:321-329, geometry.rs:340-346), so a ray
    // leaving a floor under a medium's box is inside or outside that box by the sign of the residue.  Every other query starts at
    // tmin = 0.001, where a plane a residue away from the origin is rejected on t alone.  Hence the scene-wide condition (the
    // library sets fp32_planes iff the tree holds no medium); with media, and for wrapped shapes, the f64 path below stays.
    const bool quick = S.fp32_planes != 0u && P.xf_count == 0 && !(P.flags & RT_PRIM_MEDIUM) && (P.geom == RT_GEOM_BOX || P.geom == RT_GEOM_RECT) && !want_uv;
    if (quick) {
        uint32_t axis; float plane; bool flip = (P.flags & RT_PRIM_FLIP) != 0;
        if (P.geom == RT_GEOM_RECT) {
            const DevRect& R = S.rects[P.geom_index];
            axis = R.axis; plane = R.h;
        } else {
            const DevBox& B = S.boxes[P.geom_index];
            axis = (uint32_t)(side >> 1);
            const bool upper = (side & 1) != 0;
            plane = axis == 0 ? (upper ? B.max[0] : B.min[0]) : (axis == 1 ? (upper ? B.max[1] : B.min[1]) : (upper ? B.max[2] : B.min[2]));
            flip = flip != !upper;  // the x0 / y0 / z0 sides are Flipped (outward normals)
        }
        V3 pt = ray_at(o, d, t);
        const bool ahead = t > 0.0f;
        const float s = flip ? -1.0f : 1.0f;
        h.point = v3(axis == 0 && ahead ? plane : pt.x, axis == 1 && ahead ? plane : pt.y, axis == 2 && ahead ? plane : pt.z);
        h.normal = v3(axis == 0 ? s : 0.0f, axis == 1 ? s : 0.0f, axis == 2 ? s : 0.0f);
        h.u = 0.0f; h.v = 0.0f; h.point_side = 0;
        return;
    }
    D3 lo = d3(o), ld = d3(d);
    to_local(S, P, lo, ld);
    D3 pt, n; double u, v;
    local_surface(S, P, lo, ld, time, t, side, want_uv, pt, n, u, v);
    // a medium's hit lives in the frame of the ConstantMedium wrapper: the inner wrappers move only the point
    const uint32_t n_from = (P.flags & RT_PRIM_MEDIUM) ? P.xf_outer : P.xf_count;
    for (uint32_t k = P.xf_count; k > 0; --k) {
        const rt_xform X = S.xforms[P.xf_first + k - 1];
        if (X.kind == RT_XF_TRANSLATE) pt = pt + d3(X.a, X.b, X.c);
        else { pt = rotate_cs(pt, X.kind, X.a, X.b); if (k <= n_from) n = rotate_cs(n, X.kind, X.a, X.b); }
    }
    if (P.flags & RT_PRIM_FLIP) n = n * -1.0;
    h.point = v3(pt); h.normal = v3(n); h.u = (float)u; h.v = (float)v;
    h.point_side = 0;
    const bool off_grid = S.side_rounding != 0u && plane_off_grid(P);
    if (RT_ANY_LANE(off_grid)) if (off_grid) {
        const D3 e = pt - d3(h.point);
        h.point_side = (e.x > 0.0 ? 1u : 0u) | (e.x < 0.0 ? 2u : 0u) | (e.y > 0.0 ? 4u : 0u) | (e.y < 0.0 ? 8u : 0u) | (e.z > 0.0 ? 16u : 0u) | (e.z < 0.0 ? 32u : 0u);
    }
}
RT_HD void prim_surface(const SceneView& S, uint32_t prim, V3 o, V3 d, float time, float t, int side, Hit& h) {
    prim_surface(S, S.prims[prim], prim, o, d, time, t, side, true, true, h);
}

// --------------------------------------------------------------- traversal
struct Ray {
    V3 o, d;
    float time;
};

// BvhNode::scatter_ (lib.rs:228-257) as a stackless walk over the pre-order node array:
// a node whose box the ray misses within (tmin, tmax) is skipped with its subtree, otherwise
// the walk moves to the next node in pre-order (the left child); an accepted leaf hit lowers
// tmax, and a later accepted hit replaces an earlier one.  Leaves are reached in the
// reference's left-then-right order.  Node boxes are widened by the host, so the fp32 slab
// test errs on the side of visiting.
//
// The walk runs on the library's own copy of the node array (make_walk_nodes below): 32-byte slots with the bounds
// paired per axis, the caller's leaf bits (primitive id = RT_NODE_PRIM_MASK, RT_BVH_EXACT) and two more that only say
// where the numbers of a leaf's primitive are (the arithmetic is the same with or without them):
//   (RT_BVH_EXACT)        the array slot after an exact leaf holds its exact box {min.xyz, max.xyz} for the reference's own
//                         box test at the leaf; the box in the node is widened like every other (it only gates the visit);
//   RT_NODE_PLAIN_BOX     exact leaf whose primitive is a box without wrappers or medium: the leaf's exact box is the
//                         geometry, and a second slot holds the fp32 remainders of the box's f64 coordinates (see DevBox);
//   RT_NODE_PLAIN_RECT    leaf of a Rectangle without wrappers or medium, made an exact leaf by the library: the first slot
//                         holds the rectangle's bound() (shape.rs:457-481, computed here in f64 from the rt_rect) and the
//                         plane's coordinate as value + remainder, the second the bound's remainders and the axis;
//   RT_NODE_PLAIN_SPHERE  leaf of a Sphere that is not a medium's boundary: the array slot after the leaf holds
//                         {center.x, center.y, center.z, radius} as four doubles (the leaf's skip link jumps over it).
//                         Behind Translated / Rotated wrappers the centre is the wrapped sphere's centre carried out
//                         through the wrappers: the near root along the ray in the outer frame is the near root along
//                         the transformed ray in the sphere's frame (rotations keep lengths), to a few f64 ulps.
// A leaf's word also has RT_NODE_LEAF set; the word of an inner node is the index of its first child (the next slot),
// below RT_NODE_LEAF, so that a step picks its successor from the node's two words without arithmetic.
// The slot after the last node is a sentinel (both links to itself) that a finished or idle walk may read and step on.
#define RT_NODE_PRIM_MASK 0x07FFFFFFu
#define RT_NODE_PLAIN_RECT 0x08000000u
#define RT_NODE_LEAF 0x10000000u
#define RT_NODE_PLAIN_BOX 0x40000000u
#define RT_NODE_PLAIN_SPHERE 0x20000000u
#define RT_NODE_MOVING_SPHERE (RT_NODE_PLAIN_SPHERE | RT_NODE_PLAIN_RECT)  // (a sphere leaf is no rectangle: the two bits together mark an unwrapped MovingSphere)

// A node's box is held as centre and half-widths: the two plane parameters of an axis are then
//   (c * inv - o * inv) -+ h * |inv|,   near first whatever the sign of the direction,
// so the slab test needs no minimum / maximum per axis (see node_hit).  make_node_box rounds the half-widths up so that
// [c - h, c + h] contains the box it is made from.
struct alignas(16) WalkNode { float hx, hy, cx, cy, hz, cz; uint32_t skip, prim; };
static_assert(sizeof(WalkNode) == 32, "node slot size");
// (host side) Centre and half-width of the fp32 interval [lo, hi] such that c - h <= lo and c + h >= hi hold exactly
// (evaluated in f64, where sums of two floats are exact).  An unbounded or NaN interval becomes (0, inf): every ray passes.
inline void centre_half(float lo, float hi, float& c, float& h) {
    if (!(lo - lo == 0.0f) || !(hi - hi == 0.0f)) { c = 0.0f; h = INFINITY; return; }
    c = (float)(0.5 * ((double)lo + (double)hi));
    h = (float)(0.5 * ((double)hi - (double)lo));
    if (lo > hi) return;  // an empty interval stays empty (h < 0: the near plane lies behind the far one)
    while ((double)c - (double)h > (double)lo || (double)c + (double)h < (double)hi) h = std::nextafterf(h, INFINITY);
}
inline void set_node_box(WalkNode& W, float x0, float x1, float y0, float y1, float z0, float z1) {
    centre_half(x0, x1, W.cx, W.hx); centre_half(y0, y1, W.cy, W.hy); centre_half(z0, z1, W.cz, W.hz);
}
// Positions in the node array -- the links in the nodes, the walk's place, the noted leaves -- are byte offsets
// (slot number * 32): a read from the array then needs no address arithmetic beyond its base.
#define RT_NODE_STRIDE 32u
RT_HD const WalkNode& node_at(const WalkNode* base, uint32_t offset) {  // (the array and every position in it are at least 16-byte aligned)
    return *static_cast<const WalkNode*>(__builtin_assume_aligned(reinterpret_cast<const char*>(base) + offset, 16));
}

// (host side) The inner nodes over the leaf sequence are regrouped.  The walk tests every child of a node it enters, so
// the expected number of box tests is the sum over all nodes of P(ray meets the parent's box), about
// area(parent) / area(root) for rays that cross the scene.  Leaves, their order and their boxes are fixed (the order is
// observable, see closest_hit), but which runs of consecutive leaves share an inner node is free, and an array node may
// have any number of children.  Bottom-up, for every inner node of the caller's tree: its descendants are opened up
// to a handful of "atoms" (subtrees and leaves, in order), and a dynamic programme over the runs of consecutive atoms
// picks the grouping into new inner nodes (box = union of the run) that minimises the expected tests below the node;
// a run [a, b) costs  min over its partitions into >= 2 groups of  sum_g (1 + area(g) / area(run) * cost(g)).
// Inner boxes only cull, so hits do not change.  final_scene: 55.2 box tests per segment with the caller's tree
// (median splits on random axes, lib.rs:160-200), 41.6 after merely dropping its poor inner nodes, fewer with this.
struct RegroupNode { float mn[3], mx[3]; uint32_t prim; std::vector<uint32_t> kids; };  // prim == RT_BVH_NONE: inner
// Optionally the expected number of tests is measured instead of modelled: with a sample of the rays a frame really traces
// (`SampleRays`, taken by the renderer from its queues), the share of rays that meet a box replaces the box's area, counted
// among the rays that meet the node being regrouped (its runs lie inside its box).  Rays of a path tracer start on the
// scene's surfaces and mostly end nearby, which areas -- the measure of lines through the whole scene -- know nothing about:
// final_scene needs 27.2 box tests per segment with 300 sample rays where areas give 34.0 (same leaves, same hits).
struct SampleRays {
    std::vector<float> o, inv;  // three floats per ray: origin, 1 / direction
    size_t size() const { return o.size() / 3; }
    void add(const float* origin, const float* direction) { for (int c = 0; c < 3; ++c) { o.push_back(origin[c]); inv.push_back(1.0f / direction[c]); } }
};
inline bool sample_ray_meets(const SampleRays& R, size_t k, const float* mn, const float* mx) {
    const float* o = &R.o[3 * k]; const float* iv = &R.inv[3 * k];
    float tn = 0.001f, tf = INFINITY;
    for (int c = 0; c < 3; ++c) {
        const float t0 = (mn[c] - o[c]) * iv[c], t1 = (mx[c] - o[c]) * iv[c];
        tn = std::fmax(tn, std::fmin(t0, t1)); tf = std::fmin(tf, std::fmax(t0, t1));
    }
    return tn <= tf;
}
inline void regroup_nodes(const rt_bvh_node* in, uint32_t n, std::vector<rt_bvh_node>& out, const SampleRays* rays = nullptr) {
    out.clear();
    if (n == 0) return;
    const size_t CAP = 32;  // atoms per dynamic programme (its cost grows with the fourth power; 32 measured as good as 64)
    auto area6 = [](const float* mn, const float* mx) {
        const double dx = (double)mx[0] - mn[0], dy = (double)mx[1] - mn[1], dz = (double)mx[2] - mn[2];
        const double a = 2.0 * (dx * dy + dy * dz + dz * dx);
        return (a == a && a > 0.0 && a < 1e300) ? a : 0.0;
    };
    // The measure of a box inside the node being regrouped: its area, or (with sample rays) the number of the node's rays
    // that meet it plus a little of its area, which keeps boxes no sample ray meets in proportion.
    const size_t n_rays = rays ? rays->size() : 0;
    const double root_area = n ? area6(in[0].min, in[0].max) : 0.0;
    const double prior = (n_rays && root_area > 0.0) ? 0.05 * (double)n_rays / root_area : 0.0;
    std::vector<uint32_t> cand;  // the sample rays that meet the node being regrouped
    auto measure = [&](const float* mn, const float* mx) {
        if (!n_rays) return area6(mn, mx);
        size_t hits = 0;
        for (uint32_t k : cand) hits += sample_ray_meets(*rays, k, mn, mx);
        return (double)hits + prior * area6(mn, mx);
    };
    double max_abs = 0.0;
    for (uint32_t i = 0; i < n; ++i)
        for (int c = 0; c < 3; ++c) {
            if (in[i].min[c] - in[i].min[c] == 0.0f) max_abs = std::fmax(max_abs, std::fabs((double)in[i].min[c]));
            if (in[i].max[c] - in[i].max[c] == 0.0f) max_abs = std::fmax(max_abs, std::fabs((double)in[i].max[c]));
        }
    const double pad = std::fmax(1e-6, 4e-6 * max_abs);  // new inner boxes are widened like the caller's
    auto down = [&](double x) { float f = (float)(x - pad); while ((double)f > x - pad) f = std::nextafterf(f, -INFINITY); return f; };
    auto up = [&](double x) { float f = (float)(x + pad); while ((double)f < x + pad) f = std::nextafterf(f, INFINITY); return f; };

    // the new tree, nodes appended as they are made; new_of[i] = its node for the caller's node i
    std::vector<RegroupNode> T;
    std::vector<uint32_t> new_of(n, 0);
    std::vector<double> cost(n, 0.0);                 // expected tests below caller's node i, given a ray in its box
    std::vector<std::vector<uint32_t>> atoms(n);      // caller's nodes that are the atoms of inner node i
    for (uint32_t i = n; i-- > 0;) {
        RegroupNode R;
        for (int c = 0; c < 3; ++c) { R.mn[c] = in[i].min[c]; R.mx[c] = in[i].max[c]; }
        R.prim = in[i].prim;
        if (in[i].prim != RT_BVH_NONE) { new_of[i] = (uint32_t)T.size(); T.push_back(R); continue; }
        if (n_rays) {
            cand.clear();
            for (size_t k = 0; k < n_rays; ++k) if (sample_ray_meets(*rays, k, in[i].min, in[i].max)) cand.push_back((uint32_t)k);
        }
        const double Ai = measure(in[i].min, in[i].max);
        std::vector<uint32_t>& at = atoms[i];
        for (uint32_t c = i + 1; c < in[i].skip && c < n; c = in[c].skip) {
            if (in[c].prim == RT_BVH_NONE && at.size() + atoms[c].size() <= CAP)  // an inner child is opened up while there is room
                at.insert(at.end(), atoms[c].begin(), atoms[c].end());
            else at.push_back(c);
            if (in[c].skip <= c) break;  // malformed link: stop rather than loop (the caller has validated the array)
        }
        const int k = (int)at.size();
        if (k == 0) { new_of[i] = (uint32_t)T.size(); T.push_back(R); continue; }
        // union boxes and areas of the runs [a, b)
        std::vector<double> U((size_t)k * (k + 1), 0.0);
        std::vector<float> ubox((size_t)k * (k + 1) * 6, 0.0f);
        for (int a = 0; a < k; ++a) {
            float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
            for (int b = a; b < k; ++b) {
                for (int c = 0; c < 3; ++c) { mn[c] = std::fmin(mn[c], in[at[b]].min[c]); mx[c] = std::fmax(mx[c], in[at[b]].max[c]); }
                U[(size_t)a * (k + 1) + b + 1] = measure(mn, mx);
                float* ub = &ubox[((size_t)a * (k + 1) + b + 1) * 6];
                for (int c = 0; c < 3; ++c) { ub[c] = mn[c]; ub[3 + c] = mx[c]; }
            }
        }
        // s(a, b): area-weighted expected tests inside the group [a, b): an atom's area * cost, or for a run the minimum
        // over its partitions into >= 2 groups of sum_g (area(run) + s(g)).  cut(a, b): the first group's end in the best partition
        // of the prefix ending at each position (for the walk back).
        std::vector<double> s_of((size_t)k * (k + 1), 0.0);
        std::vector<std::vector<int>> pred((size_t)k * (k + 1));
        auto S = [&](int a, int b) -> double& { return s_of[(size_t)a * (k + 1) + b]; };
        for (int a = 0; a < k; ++a) S(a, a + 1) = measure(in[at[a]].min, in[at[a]].max) * cost[at[a]];
        auto partition = [&](int a, int b, double parent_area, bool need_two, std::vector<int>& pr) {
            const int m = b - a;
            std::vector<double> G(m + 1, INFINITY);
            pr.assign(m + 1, 0);
            G[0] = 0.0;
            for (int j = 1; j <= m; ++j)
                for (int i2 = 0; i2 < j; ++i2) {
                    if (need_two && i2 == 0 && j == m) continue;
                    const double v = G[i2] + parent_area + S(a + i2, a + j);
                    if (v < G[j]) { G[j] = v; pr[j] = i2; }
                }
            return G[m];
        };
        for (int len = 2; len <= k; ++len)
            for (int a = 0; a + len <= k; ++a) {
                const int b = a + len;
                S(a, b) = partition(a, b, U[(size_t)a * (k + 1) + b], true, pred[(size_t)a * (k + 1) + b]);
            }
        std::vector<int> top;
        const double total = k > 1 ? partition(0, k, Ai, false, top) : Ai + S(0, 1);
        cost[i] = Ai > 0.0 ? total / Ai : (double)k;
        // build the new nodes of this subtree: groups of a partition, recursively
        struct Build {
            static uint32_t group(std::vector<RegroupNode>& T, const std::vector<uint32_t>& at, const std::vector<uint32_t>& new_of, int k,
                                  const std::vector<std::vector<int>>& pred, const std::vector<float>& ubox, int a, int b, const RegroupNode& within,
                                  const std::function<float(double)>& down, const std::function<float(double)>& up) {
                if (b - a == 1) return new_of[at[a]];
                RegroupNode G;
                const float* ub = &ubox[((size_t)a * (k + 1) + b) * 6];
                // the union of the run, widened, and cut back to the enclosing node's box (which holds all of the run's leaves as well)
                for (int c = 0; c < 3; ++c) { G.mn[c] = std::fmax(down((double)ub[c]), within.mn[c]); G.mx[c] = std::fmin(up((double)ub[3 + c]), within.mx[c]); }
                G.prim = RT_BVH_NONE;
                const std::vector<int>& pr = pred[(size_t)a * (k + 1) + b];
                std::vector<std::pair<int, int>> parts;
                for (int j = b - a; j > 0; j = pr[j]) parts.push_back({a + pr[j], a + j});
                for (size_t q = parts.size(); q-- > 0;) G.kids.push_back(group(T, at, new_of, k, pred, ubox, parts[q].first, parts[q].second, within, down, up));
                T.push_back(G);
                return (uint32_t)T.size() - 1;
            }
        };
        const std::function<float(double)> fdown = down, fup = up;
        if (k > 1) {
            std::vector<std::pair<int, int>> parts;
            for (int j = k; j > 0; j = top[j]) parts.push_back({top[j], j});
            for (size_t q = parts.size(); q-- > 0;) R.kids.push_back(Build::group(T, at, new_of, k, pred, ubox, parts[q].first, parts[q].second, R, fdown, fup));
        } else R.kids.push_back(new_of[at[0]]);
        new_of[i] = (uint32_t)T.size();
        T.push_back(R);
    }
    // emit in pre-order with an explicit stack; skip = index after the subtree
    struct Frame { uint32_t node; size_t next_kid; uint32_t at; };
    std::vector<Frame> stack;
    auto push = [&](uint32_t t) {
        rt_bvh_node N{};
        for (int c = 0; c < 3; ++c) { N.min[c] = T[t].mn[c]; N.max[c] = T[t].mx[c]; }
        N.prim = T[t].prim; N.skip = 0;
        out.push_back(N);
        stack.push_back({t, 0, (uint32_t)out.size() - 1});
    };
    push(new_of[0]);
    while (!stack.empty()) {
        Frame& f = stack.back();
        if (f.next_kid == T[f.node].kids.size()) { out[f.at].skip = (uint32_t)out.size(); stack.pop_back(); continue; }
        const uint32_t c = T[f.node].kids[f.next_kid++];
        push(c);
    }
}

// (host side) The walk's node array for a scene description: inner nodes regrouped, payload slots inserted, skip links
// remapped, leaves annotated, sentinel appended.  nodes.size() - 1 is the SceneView's n_bvh_nodes.
inline void make_walk_nodes(const rt_scene_desc* D, std::vector<WalkNode>& nodes, const SampleRays* rays = nullptr) {
    std::vector<rt_bvh_node> tree;
    regroup_nodes(D->bvh, D->n_bvh_nodes, tree, rays);
    const uint32_t n_in = (uint32_t)tree.size();
    std::vector<uint32_t> new_index(n_in + 1);
    auto plain_sphere = [&](const rt_bvh_node& N) {
        if (N.prim == RT_BVH_NONE) return false;
        const rt_prim& Pp = D->prims[N.prim & RT_NODE_PRIM_MASK];
        return Pp.geom == RT_GEOM_SPHERE && !(Pp.flags & (RT_PRIM_MEDIUM | RT_PRIM_BOTH_ROOTS));  // (the far-root variant takes the general test)
    };
    auto moving_sphere = [&](const rt_bvh_node& N) {  // (unwrapped: a wrapped one takes the general test)
        if (N.prim == RT_BVH_NONE) return false;
        const rt_prim& Pp = D->prims[N.prim & RT_NODE_PRIM_MASK];
        return Pp.geom == RT_GEOM_MOVING_SPHERE && Pp.xf_count == 0 && !(Pp.flags & (RT_PRIM_MEDIUM | RT_PRIM_BOTH_ROOTS));
    };
    auto plain_box = [&](const rt_bvh_node& N) {  // exact leaf of an unwrapped box whose box is the leaf's box (rounded to fp32)
        if (N.prim == RT_BVH_NONE || !(N.prim & RT_BVH_EXACT)) return false;
        const rt_prim& Pp = D->prims[N.prim & RT_NODE_PRIM_MASK];
        if (Pp.geom != RT_GEOM_BOX || Pp.xf_count != 0 || (Pp.flags & RT_PRIM_MEDIUM)) return false;
        const rt_box& B = D->boxes[Pp.geom_index];
        bool same = true;
        for (int c = 0; c < 3; ++c) same = same && (float)B.min[c] == N.min[c] && (float)B.max[c] == N.max[c];
        return same;
    };
    auto plain_rect = [&](const rt_bvh_node& N) {
        if (N.prim == RT_BVH_NONE) return false;
        const rt_prim& Pp = D->prims[N.prim & RT_NODE_PRIM_MASK];
        return Pp.geom == RT_GEOM_RECT && Pp.xf_count == 0 && !(Pp.flags & RT_PRIM_MEDIUM);
    };
    auto exact_leaf = [&](const rt_bvh_node& N) { return N.prim != RT_BVH_NONE && ((N.prim & RT_BVH_EXACT) != 0 || plain_rect(N)); };
    auto payload_slots = [&](const rt_bvh_node& N) { return plain_sphere(N) ? 1u : ((moving_sphere(N) || plain_box(N) || plain_rect(N)) ? 2u : (exact_leaf(N) ? 1u : 0u)); };
    // exact leaves keep their exact box in a payload slot; the box in the node itself, which only decides whether the walk
    // visits the leaf, is widened like every other node box (the fp32 slab test is accurate to ulps of origin / direction)
    double max_abs = 0.0;
    for (uint32_t i = 0; i < n_in; ++i)
        for (int c = 0; c < 3; ++c) {
            if (tree[i].min[c] - tree[i].min[c] == 0.0f) max_abs = std::fmax(max_abs, std::fabs((double)tree[i].min[c]));
            if (tree[i].max[c] - tree[i].max[c] == 0.0f) max_abs = std::fmax(max_abs, std::fabs((double)tree[i].max[c]));
        }
    const double pad = std::fmax(1e-6, 4e-6 * max_abs);
    uint32_t at = 0;
    for (uint32_t i = 0; i < n_in; ++i) { new_index[i] = at; at += 1u + payload_slots(tree[i]); }
    new_index[n_in] = at;
    nodes.assign((size_t)at + 1, WalkNode{0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, at * RT_NODE_STRIDE, at * RT_NODE_STRIDE});
    for (uint32_t i = 0; i < n_in; ++i) {
        const rt_bvh_node& N = tree[i];
        WalkNode W{0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, new_index[N.skip] * RT_NODE_STRIDE, (new_index[i] + 1u) * RT_NODE_STRIDE};
        set_node_box(W, N.min[0], N.max[0], N.min[1], N.max[1], N.min[2], N.max[2]);
        if (N.prim != RT_BVH_NONE) {
            W.prim = RT_NODE_LEAF | (N.prim & RT_BVH_EXACT) | (N.prim & RT_NODE_PRIM_MASK);
            const rt_prim& Pp = D->prims[N.prim & RT_NODE_PRIM_MASK];
            if (plain_sphere(N)) {
                W.prim |= RT_NODE_PLAIN_SPHERE;
                const rt_sphere& sp = D->spheres[Pp.geom_index];
                D3 c = d3(sp.center);
                for (uint32_t k = Pp.xf_count; k > 0; --k) {  // innermost wrapper first, as prim_surface carries a point out
                    const rt_xform& X = D->xforms[Pp.xf_first + k - 1];
                    if (X.kind == RT_XF_TRANSLATE) c = c + d3(X.a, X.b, X.c);
                    else c = rotate_cs(c, X.kind, X.a, X.b);
                }
                const double payload[4] = {c.x, c.y, c.z, sp.radius};
                static_assert(sizeof(payload) == sizeof(WalkNode), "payload slot size");
                std::memcpy(&nodes[new_index[i] + 1], payload, sizeof(payload));
            } else if (moving_sphere(N)) {
                // centre at the start time and radius as a plain sphere's, then the motion per unit of time and the start time
                W.prim |= RT_NODE_MOVING_SPHERE;
                const rt_sphere& sp = D->spheres[Pp.geom_index];
                const double payload[8] = {sp.center[0], sp.center[1], sp.center[2], sp.radius, sp.motion[0], sp.motion[1], sp.motion[2], sp.start_time};
                static_assert(sizeof(payload) == 2 * sizeof(WalkNode), "payload slot size");
                std::memcpy(&nodes[new_index[i] + 1], payload, sizeof(payload));
            } else if (plain_rect(N)) {
                // Shape::bound of a Rectangle (shape.rs:457-481): its extents, and the plane padded by 0.0001 on its axis
                W.prim |= RT_BVH_EXACT | RT_NODE_PLAIN_RECT;
                const rt_rect& R = D->rects[Pp.geom_index];
                int p1, p2, oo; axis_perm(R.axis, p1, p2, oo);
                double mn[3], mx[3];
                mn[p1] = R.x0; mn[p2] = R.y0; mn[oo] = R.height - 0.0001;
                mx[p1] = R.x1; mx[p2] = R.y1; mx[oo] = R.height + 0.0001;
                float exact[8], rem[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
                for (int c = 0; c < 3; ++c) { split_hi_lo(mn[c], exact[c], rem[c]); split_hi_lo(mx[c], exact[3 + c], rem[3 + c]); }
                split_hi_lo(R.height, exact[6], exact[7]);
                const uint32_t axis = R.axis;
                std::memcpy(&rem[6], &axis, sizeof(axis));
                std::memcpy(&nodes[new_index[i] + 1], exact, sizeof(exact));
                std::memcpy(&nodes[new_index[i] + 2], rem, sizeof(rem));
                auto down = [&](float x) { float f = (float)((double)x - pad); while ((double)f > (double)x - pad) f = std::nextafterf(f, -INFINITY); return f; };
                auto up = [&](float x) { float f = (float)((double)x + pad); while ((double)f < (double)x + pad) f = std::nextafterf(f, INFINITY); return f; };
                set_node_box(W, down(exact[0]), up(exact[3]), down(exact[1]), up(exact[4]), down(exact[2]), up(exact[5]));
            } else if (exact_leaf(N)) {
                const float exact[8] = {N.min[0], N.min[1], N.min[2], N.max[0], N.max[1], N.max[2], 0.0f, 0.0f};
                std::memcpy(&nodes[new_index[i] + 1], exact, sizeof(exact));
                if (plain_box(N)) {
                    W.prim |= RT_NODE_PLAIN_BOX;
                    const rt_box& B = D->boxes[Pp.geom_index];
                    float rem[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f}, hi;
                    for (int c = 0; c < 3; ++c) { split_hi_lo(B.min[c], hi, rem[c]); split_hi_lo(B.max[c], hi, rem[3 + c]); }
                    std::memcpy(&nodes[new_index[i] + 2], rem, sizeof(rem));
                }
                auto down = [&](float x) { float f = (float)((double)x - pad); while ((double)f > (double)x - pad) f = std::nextafterf(f, -INFINITY); return f; };
                auto up = [&](float x) { float f = (float)((double)x + pad); while ((double)f < (double)x + pad) f = std::nextafterf(f, INFINITY); return f; };
                set_node_box(W, down(N.min[0]), up(N.max[0]), down(N.min[1]), up(N.max[1]), down(N.min[2]), up(N.max[2]));
            }
        }
        nodes[new_index[i]] = W;
    }
}

struct Walk {
    uint32_t i;     // next node to look at (byte offset in the node array, like every node position)
    uint32_t leaf_node;  // the node the last walk_step stood on
    float tmax;     // shrinking upper end of the ray interval
    float t;        // accepted hit so far
    uint32_t prim;
    int side;
    bool found;
};
RT_HD void walk_init(Walk& w, float tmax) { w.i = 0; w.leaf_node = 0; w.tmax = tmax; w.t = 0.0f; w.prim = 0; w.side = 0; w.found = false; }

// Slab test of a node with the ray given as (1 / direction, origin / direction).  With the box as centre and half-widths
// the parameters of the two planes of an axis are m -+ h * |inv|, m = c * inv - o * inv: the near one first for either
// sign of the direction.  On the GPU that is two fused multiply-adds for the three m (one of them two-wide) and one two-wide
// one per axis for (near, far); then a three-way maximum and minimum.
#if defined(__HIP_DEVICE_COMPILE__)
typedef float F2 __attribute__((ext_vector_type(2)));
RT_HD F2 fma2(float a0, float a1, float b0, float b1, float c0, float c1) { const F2 a = {a0, a1}, b = {b0, b1}, c = {c0, c1}; return __builtin_elementwise_fma(a, b, c); }
#else
struct F2 { float x, y; };
RT_HD F2 fma2(float a0, float a1, float b0, float b1, float c0, float c1) { F2 r; r.x = fmaf(a0, b0, c0); r.y = fmaf(a1, b1, c1); return r; }
#endif
// (near, far) of the three axes and the comparison; `ainv` = |inv|
RT_HD bool slab_hit(float cx, float cy, float cz, float hx, float hy, float hz, V3 inv, V3 ainv, V3 oi, float tmin, float tmax) {
    const F2 mxy = fma2(cx, cy, inv.x, inv.y, -oi.x, -oi.y);
    const float mz = fmaf(cz, inv.z, -oi.z);
    const F2 x = fma2(-hx, hx, ainv.x, ainv.x, mxy.x, mxy.x), y = fma2(-hy, hy, ainv.y, ainv.y, mxy.y, mxy.y), z = fma2(-hz, hz, ainv.z, ainv.z, mz, mz);
    const float tn = fmaxf(fmaxf(fmaxf(x.x, y.x), z.x), tmin);
    const float tf = fminf(fminf(fminf(x.y, y.y), z.y), tmax);
    return tn <= tf;
}
RT_HD V3 abs3(V3 a) { return v3(fabsf(a.x), fabsf(a.y), fabsf(a.z)); }
RT_HD bool node_hit(const WalkNode& N, V3 inv, V3 oi, float tmin, float tmax) {
    return slab_hit(N.cx, N.cy, N.cz, N.hx, N.hy, N.hz, inv, abs3(inv), oi, tmin, tmax);
}

// One node step.  Returns the leaf's prim field when the ray now stands on a leaf whose primitive has to be
// tried (w.i already points past it), RT_BVH_NONE otherwise.  Requires w.i to be before the sentinel.
RT_HD uint32_t walk_step(const SceneView& S, V3 inv, V3 oi, float tmin, Walk& w) {
    const WalkNode N = node_at(S.bvh, w.i);
    RT_COUNT(nodes);
    if (!node_hit(N, inv, oi, tmin, fminf(w.tmax, RT_WALK_FAR))) { w.i = N.skip; return RT_BVH_NONE; }
    w.leaf_node = w.i;
    if (N.prim < RT_NODE_LEAF) { w.i = N.prim; return RT_BVH_NONE; }  // inner: on to its first child
    w.i = N.skip;  // a leaf's skip link also jumps over its payload slot, if any
    return N.prim;
}
// The same step without control flow, for lanes in lockstep, with up to two leaves left pending: a leaf the walk meets
// is noted (its node index, oldest in p0) and the walk goes on; the primitive tests are made later, oldest first.
// That changes no result: the tests still run in leaf order against the bound as it is then, a walk that goes on with a
// bound not yet lowered only visits nodes it might have skipped, and a leaf reached that way fails its test (its
// primitive lies inside its box, beyond the bound; exact leaves repeat their box test when they are tried).
// A walk with two leaves pending, or one that has left the tree (w.i on the sentinel slot), stays as it is.
#if defined(__HIPCC__)
// m0 / m1: the lanes of the wave whose p0 / p1 hold a leaf (p0 != RT_BVH_NONE, p1 != RT_BVH_NONE), kept as wave masks by
// the caller over a run of steps.  All of a step's conditions are formed as wave masks from its two comparisons
// (mask arithmetic runs on the scalar unit), so a step spends its vector instructions on the slab test and four selects.
// LDS_AT_ZERO: the node array stands in LDS from address 0 on, so a node's position is its address (the caller has checked
// that the kernel has no LDS of its own in front of the dynamic array).
template <bool LDS_AT_ZERO>
__device__ __forceinline__ void walk_step_flat(const SceneView& S, V3 inv, V3 ainv, V3 oi, float tmin, Walk& w, uint32_t& p0, uint32_t& p1, unsigned long long& m0, unsigned long long& m1) {
    typedef __attribute__((address_space(3))) const WalkNode* LdsNode;
    WalkNode N;
    if (LDS_AT_ZERO) { const LdsNode at = reinterpret_cast<LdsNode>((size_t)w.i); N.hx = at->hx; N.hy = at->hy; N.cx = at->cx; N.cy = at->cy; N.hz = at->hz; N.cz = at->cz; N.skip = at->skip; N.prim = at->prim; }
    else N = node_at(S.bvh, w.i);
    const bool in_box = slab_hit(N.cx, N.cy, N.cz, N.hx, N.hy, N.hz, inv, ainv, oi, tmin, w.tmax);
    const unsigned long long hit_m = __builtin_amdgcn_ballot_w64(in_box) & ~m1;
    const unsigned long long leaf_m = __builtin_amdgcn_ballot_w64(N.prim >= RT_NODE_LEAF) & hit_m;
    const uint32_t next = __builtin_amdgcn_inverse_ballot_w64(hit_m & ~leaf_m) ? N.prim : N.skip;  // (on the sentinel both are the sentinel)
    p1 = __builtin_amdgcn_inverse_ballot_w64(leaf_m & m0) ? w.i : p1;
    p0 = __builtin_amdgcn_inverse_ballot_w64(leaf_m & ~m0) ? w.i : p0;
    w.i = __builtin_amdgcn_inverse_ballot_w64(m1) ? w.i : next;
    m1 |= leaf_m & m0;
    m0 |= leaf_m;
}
#endif

// The primitive test of the leaf at node w.leaf_node: Object::scatter's intersect (lib.rs:284-285) behind the
// leaf's own bounding-box test (lib.rs:236-240).  walk_leaf_fast handles the leaves that need nothing but the
// node array (and no random draw) and returns false for the rest, which walk_leaf_generic takes.
RT_HD void walk_accept(Walk& w, float t, uint32_t prim, int side) { w.tmax = t; w.t = t; w.prim = prim; w.side = side; w.found = true; }
// `inv_capped`: 1 / direction as the walk holds it (see walk_cap); only the box leaves use it.
RT_HD bool walk_leaf_fast(const SceneView& S, const Ray& r, V3 inv_capped, float tmin, uint32_t leaf, Walk& w) {
    const uint32_t prim = leaf & RT_NODE_PRIM_MASK;
    if (leaf & RT_NODE_PLAIN_SPHERE) {
        RT_COUNT(leaves); RT_COUNT(prim_tests);
        const double* c = reinterpret_cast<const double*>(&node_at(S.bvh, w.leaf_node + RT_NODE_STRIDE));
        rt_sphere sp;
        sp.center[0] = c[0]; sp.center[1] = c[1]; sp.center[2] = c[2]; sp.radius = c[3];
        const bool moving = (leaf & RT_NODE_PLAIN_RECT) != 0u;  // RT_NODE_MOVING_SPHERE: the second payload slot holds motion and start time; needs r.time
        if (moving) { sp.motion[0] = c[4]; sp.motion[1] = c[5]; sp.motion[2] = c[6]; sp.start_time = c[7]; }
        double t64;
        if (sphere_root(sp, moving, d3(r.o), d3(r.d), r.time, t64) && t64 > (double)tmin && t64 < (double)w.tmax) walk_accept(w, (float)t64, prim, 0);
        return true;
    }
    if (!(leaf & RT_BVH_EXACT)) return false;
    // BoundingBox::hit (shape.rs:580-599) on the primitive's own box: (bound - origin) over the direction,
    // swap for negative directions, NaN-ignoring min/max, miss when tmax <= tmin.
    const bool plain = (leaf & (RT_NODE_PLAIN_BOX | RT_NODE_PLAIN_RECT)) != 0;  // two payload slots: the second holds the remainders
    const float* exact = reinterpret_cast<const float*>(&node_at(S.bvh, w.leaf_node + RT_NODE_STRIDE));
    const float* rem = reinterpret_cast<const float*>(&node_at(S.bvh, w.leaf_node + (plain ? 2u : 1u) * RT_NODE_STRIDE));  // the remainders; any slot when unused
    DevBox B;
    for (int c = 0; c < 3; ++c) { B.min[c] = exact[c]; B.max[c] = exact[3 + c]; B.min_lo[c] = plain ? rem[c] : 0.0f; B.max_lo[c] = plain ? rem[3 + c] : 0.0f; }
    float q[6];
    const V3 inv = walk_uncap(inv_capped);  // the reference's quotients: +-inf and NaN where a direction component is zero
    box_quotients(B, r.o, inv, q);
    float lo = tmin, hi = w.tmax;
    lo = fmaxf(r.d.x < 0.0f ? q[1] : q[0], lo); hi = fminf(r.d.x < 0.0f ? q[0] : q[1], hi);
    bool enter = !(hi <= lo);
    lo = fmaxf(r.d.y < 0.0f ? q[3] : q[2], lo); hi = fminf(r.d.y < 0.0f ? q[2] : q[3], hi);
    enter = enter && !(hi <= lo);
    lo = fmaxf(r.d.z < 0.0f ? q[5] : q[4], lo); hi = fminf(r.d.z < 0.0f ? q[4] : q[5], hi);
    enter = enter && !(hi <= lo);
    if (!enter) { RT_COUNT(leaves); return true; }
    if (!plain) return false;
    if (leaf & RT_NODE_PLAIN_RECT) {  // Rectangle::intersect (shape.rs:309-332) from the leaf's own numbers
        RT_COUNT(leaves); RT_COUNT(prim_tests);
        uint32_t axis; memcpy(&axis, &rem[6], sizeof(axis));
        int p1, p2, oo; axis_perm(axis, p1, p2, oo);
        (void)oo;
        float t;
        if (rect_t<float, V3>(axis, exact[p1], exact[p2], exact[3 + p1], exact[3 + p2], exact[6], exact[7], r.o, r.d, tmin, w.tmax, t)) walk_accept(w, t, prim, 0);
        return true;
    }
    RT_COUNT(leaves); RT_COUNT(prim_tests);
    float t; int side;  // the leaf's box is the box: same arithmetic as box_t<float>, without fetching it
    if (box_sides_q(B, r.o, r.d, q, tmin, w.tmax, t, side)) walk_accept(w, t, prim, side);
    return true;
}
// Everything else: the general Intersection::intersect.  (An exact leaf arrives here only after walk_leaf_fast
// has let the ray into its box.)
RT_HD void walk_leaf_generic(const SceneView& S, const Ray& r, float tmin, const DrawKey& key, uint32_t leaf, Walk& w) {
    RT_COUNT(leaves); RT_COUNT(prim_tests); RT_COUNT(generic_leaves);
    float t; int side;
    if (prim_t(S, leaf & RT_NODE_PRIM_MASK, r.o, r.d, tmin, w.tmax, r.time, key, t, side)) walk_accept(w, t, leaf & RT_NODE_PRIM_MASK, side);
}

// The whole walk for one ray.  Two nested loops on purpose: the inner loop only steps over nodes until it
// stands on a leaf, the outer loop runs the (much longer) primitive test, so the lanes of a wave run their
// primitive tests together, once per round.
// GENERIC = false: for trees none of whose leaves takes the general test (the caller has checked); the kernels built that way
// need far fewer registers.
template <bool GENERIC = true>
RT_HD bool closest_hit(const SceneView& S, const Ray& r, float tmin, float tmax, const DrawKey& key, float& best_t, uint32_t& best_prim, int& best_side) {
    if (has_negative_zero(r.d)) return false;
    const V3 inv_c = walk_cap(ray_inv(r.d));
    const V3 oi = v3(r.o.x * inv_c.x, r.o.y * inv_c.y, r.o.z * inv_c.z);
    Walk w; walk_init(w, tmax);
    const uint32_t n = S.n_bvh_nodes * RT_NODE_STRIDE;
    for (;;) {
        uint32_t leaf = RT_BVH_NONE;
        while (w.i < n && leaf == RT_BVH_NONE) leaf = walk_step(S, inv_c, oi, tmin, w);
        if (leaf == RT_BVH_NONE) break;
        if (!walk_leaf_fast(S, r, inv_c, tmin, leaf, w) && GENERIC) walk_leaf_generic(S, r, tmin, key, leaf, w);
    }
    best_t = w.t; best_prim = w.prim; best_side = w.side;
    return w.found;
}

// ---------------------------------------------------------------- textures
// noise 0.6.0 `Perlin::get` for [f64;3] points (third party, restated from the
// crate's published algorithm; not verifiable offline): eight surflets
// attn^4 * (d . grad), attn = 1 - |d|^2, scaled and clamped to [-1,1].
RT_HD V3 perlin_grad3(uint32_t index) {
    const float d = 0.70710678118654752f, e = 0.57735026918962576f;
    uint32_t i = index & 31u;
    if (i >= 24u) return v3((i & 1u) ? -e : e, (i & 2u) ? -e : e, (i & 4u) ? -e : e);
    if (i >= 12u) i -= 12u;
    float a = (i & 1u) ? -d : d, b = (i & 2u) ? -d : d;
    if (i < 4u) return v3(a, b, 0.0f);
    if (i < 8u) return v3(a, 0.0f, b);
    return v3(0.0f, a, b);
}
#if defined(__HIP_DEVICE_COMPILE__)
// perlin_grad3 as a table (the same floats): one 16-byte load in place of a dozen comparisons and selects per surflet.
#define RT_GRAD_D 0.70710678118654752f
#define RT_GRAD_E 0.57735026918962576f
#define D RT_GRAD_D
#define E RT_GRAD_E
__constant__ __attribute__((aligned(16))) float rt_perlin_grads[32][4] = {
    {D, D, 0.0f, 0.0f}, {-D, D, 0.0f, 0.0f}, {D, -D, 0.0f, 0.0f}, {-D, -D, 0.0f, 0.0f},
    {D, 0.0f, D, 0.0f}, {-D, 0.0f, D, 0.0f}, {D, 0.0f, -D, 0.0f}, {-D, 0.0f, -D, 0.0f},
    {0.0f, D, D, 0.0f}, {0.0f, -D, D, 0.0f}, {0.0f, D, -D, 0.0f}, {0.0f, -D, -D, 0.0f},
    {D, D, 0.0f, 0.0f}, {-D, D, 0.0f, 0.0f}, {D, -D, 0.0f, 0.0f}, {-D, -D, 0.0f, 0.0f},
    {D, 0.0f, D, 0.0f}, {-D, 0.0f, D, 0.0f}, {D, 0.0f, -D, 0.0f}, {-D, 0.0f, -D, 0.0f},
    {0.0f, D, D, 0.0f}, {0.0f, -D, D, 0.0f}, {0.0f, D, -D, 0.0f}, {0.0f, -D, -D, 0.0f},
    {E, E, E, 0.0f}, {-E, E, E, 0.0f}, {E, -E, E, 0.0f}, {-E, -E, E, 0.0f},
    {E, E, -E, 0.0f}, {-E, E, -E, 0.0f}, {E, -E, -E, 0.0f}, {-E, -E, -E, 0.0f}};
#undef D
#undef E
__device__ __forceinline__ V3 perlin_grad3_at(uint32_t index) {
    const float4 g = *reinterpret_cast<const float4*>(rt_perlin_grads[index & 31u]);
    return v3(g.x, g.y, g.z);
}
#else
RT_HD V3 perlin_grad3_at(uint32_t index) { return perlin_grad3(index); }
#endif
RT_HD float perlin3(const uint8_t* perm, V3 p) {
    float fx = floorf(p.x), fy = floorf(p.y), fz = floorf(p.z);
    int ix = (int)fx, iy = (int)fy, iz = (int)fz;
    float dx = p.x - fx, dy = p.y - fy, dz = p.z - fz;
    // the first two levels of the corners' hashes are shared: 2 + 4 + 8 look-ups instead of 3 x 8
    uint32_t hx[2], hxy[4];
#pragma unroll
    for (int c = 0; c < 2; ++c) hx[c] = perm[(uint32_t)(ix + c) & 255u];
#pragma unroll
    for (int c = 0; c < 4; ++c) hxy[c] = perm[hx[c & 1] ^ ((uint32_t)(iy + (c >> 1)) & 255u)];
    float total = 0.0f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        int cx = c & 1, cy = (c >> 1) & 1, cz = c >> 2;
        V3 dd = v3(dx - (float)cx, dy - (float)cy, dz - (float)cz);
        float attn = 1.0f - dot(dd, dd);
        if (attn > 0.0f) {
            const uint32_t hsh = perm[hxy[c & 3] ^ ((uint32_t)(iz + cz) & 255u)];
            float a2 = attn * attn;
            total = fmaf(a2 * a2, dot(dd, perlin_grad3_at(hsh)), total);
        }
    }
    float r = total * 3.8898553255531074f;
    return fminf(fmaxf(r, -1.0f), 1.0f);
}

// The other two noise functions a noise texture can name (rt_abi.h: RT_NOISE_*), after the noise crate's `Fbm` and
// `Turbulence` (third party, restated like Perlin; no reference example uses them).  Octave i of an Fbm draws on the
// permutation table i places behind the function's first.
RT_HD float fbm3(const uint8_t* perm, uint32_t octaves, float lacunarity, float persistence, V3 p) {
    float result = 0.0f, amplitude = 1.0f;
    for (uint32_t i = 0; i < octaves; ++i) {
        result = fmaf(perlin3(perm + 256u * i, p), amplitude, result);
        amplitude *= persistence;
        p = p * lacunarity;
    }
    return result;
}
// The source Perlin at the point displaced by three Fbm functions (tables `perm`, `perm + 256`, `perm + 512` onwards),
// each looked up at the point shifted by the crate's fixed offsets.
RT_HD float turbulence3(const uint8_t* perm, const uint8_t* source, uint32_t roughness, float lacunarity, float persistence, float power, V3 p) {
    const float u = 1.0f / 65536.0f;
    const float dx = fbm3(perm, roughness, lacunarity, persistence, v3(p.x + 12414.0f * u, p.y + 65124.0f * u, p.z + 31337.0f * u));
    const float dy = fbm3(perm + 256u, roughness, lacunarity, persistence, v3(p.x + 26519.0f * u, p.y + 18128.0f * u, p.z + 60493.0f * u));
    const float dz = fbm3(perm + 512u, roughness, lacunarity, persistence, v3(p.x + 53820.0f * u, p.y + 11213.0f * u, p.z + 44845.0f * u));
    return perlin3(source, v3(fmaf(dx, power, p.x), fmaf(dy, power, p.y), fmaf(dz, power, p.z)));
}

// Checkered (textures.rs:49-55): `sin(10 x) * sin(10 y) * sin(10 z) < 0` picks the odd texture.  Only the signs count:
// sin(theta) < 0 iff floor(theta / pi) is odd, so the product is negative iff an odd number of the three floor(10 c / pi)
// are odd -- unless a factor is zero or NaN (a coordinate that is 0, infinite or NaN), when the product is not below zero.
// Three multiplications and six floors in f64 (10 c is exact there for an fp32 c; the quotient errs by 2e-16 of itself,
// far less than the fp32 rounding of the point it is taken at) in place of three sines with their argument reductions.
RT_HD bool floor_is_odd(double u) { const double k = floor(u), h = 0.5 * k; return h != floor(h); }
RT_HD bool checker_odd(V3 p) {
    const double ten_over_pi = 3.1830988618379067;
    const bool plain = fabsf(p.x) > 0.0f && fabsf(p.x) < RT_INF_F && fabsf(p.y) > 0.0f && fabsf(p.y) < RT_INF_F && fabsf(p.z) > 0.0f && fabsf(p.z) < RT_INF_F;
    const bool odd = (floor_is_odd((double)p.x * ten_over_pi) != floor_is_odd((double)p.y * ten_over_pi)) != floor_is_odd((double)p.z * ten_over_pi);
    return plain && odd;
}

// Texture::color_at (textures.rs:46-78).
RT_HD V3 texture_color(const SceneView& S, uint32_t tex, float u, float v, V3 p) {
    rt_texture T = S.textures[tex];
    if (T.kind == RT_TEX_CHECKERED) {
        const bool odd = checker_odd(p);
        for (uint32_t depth = 0; T.kind == RT_TEX_CHECKERED && depth < RT_MAX_TEXTURE_NESTING; ++depth) T = S.textures[odd ? T.odd : T.even];
    }
    if (T.kind == RT_TEX_NOISE) {
        const V3 q = v3(p.x * T.scale, p.y * T.scale, p.z * T.scale);
        const uint8_t* table = S.perm + 256u * T.resource;
        const uint32_t fn = T.odd & 255u, octaves = (T.odd >> 8) & 255u;
        float n;
        if (fn == RT_NOISE_PERLIN) n = perlin3(table, q);
        else if (fn == RT_NOISE_FBM) n = fbm3(table, octaves, T.color[0], T.color[1], q);
        else n = turbulence3(table, S.perm + 256u * T.even, octaves, T.color[0], T.color[1], T.color[2], q);
        return v3(n, n, n);  // may be negative: the reference does not clamp
    }
    if (T.kind == RT_TEX_IMAGE) {
        const rt_image I = S.images[T.resource];
        float uu = fminf(fmaxf(u, 0.0f), 1.0f);
        float vv = 1.0f - fminf(fmaxf(v, 0.0f), 1.0f);
        uint32_t i = (uint32_t)(uu * (float)I.width), j = (uint32_t)(vv * (float)I.height);
        i = i < I.width - 1 ? i : I.width - 1;
        j = j < I.height - 1 ? j : I.height - 1;
        const uint8_t* px = S.texels + I.texel_offset + 3ull * ((uint64_t)j * I.width + i);
        return v3((float)px[0] / 255.0f, (float)px[1] / 255.0f, (float)px[2] / 255.0f);
    }
    return v3(T.color[0], T.color[1], T.color[2]);
}

// --------------------------------------------------------------- materials
// OnUnitSphere (geometry.rs:372-382): r = 1 - z^2 without the square root, then normalised.
// The sampler piles probability up at the poles z = +-1, where r -> 0; the normalisation runs in
// f64 (on fp32 cos/sin) so that the z component keeps its distance from +-1, which Lambertian
// scattering off a z-facing surface depends on.
RT_HD D3 random_unit_vector_d(float u_phi, float u_z) {
    float sn, cs;
    sincos_turns(u_phi, sn, cs);
    double z = 2.0 * (double)u_z - 1.0;
    double r = 1.0 - z * z;
    double x = r * (double)cs, y = r * (double)sn;
    double inv = rsqrt_near(x * x + y * y + z * z);  // >= 3/4 of 1: never zero
    return d3(x * inv, y * inv, z * inv);
}
RT_HD V3 random_unit_vector(float u_phi, float u_z) { return v3(random_unit_vector_d(u_phi, u_z)); }
RT_HD V3 reflect(V3 v, V3 n) { return normed(v - n * (2.0f * dot(v, n))); }  // materials.rs:132
RT_HD V3 refract(V3 v, V3 n, float eta) {                                      // materials.rs:136
    float cos_theta = -dot(v, n);
    V3 out_par = (v + n * cos_theta) * eta;
    // the reference's f64 argument cannot go negative here in practice (refract is only called below the
    // critical angle); fp32 rounding can, hence the clamp
    V3 out_perp = n * -(sqrtf(fmaxf(1.0f - dot(out_par, out_par), 0.0f)));
    return out_par + out_perp;
}
RT_HD float schlick(float cosine, float ri) {  // materials.rs:143
    float r0 = (1.0f - ri) / (1.0f + ri);
    r0 = r0 * r0;
    float m = 1.0f - cosine, m2 = m * m;
    return r0 + (1.0f - r0) * (m2 * m2 * m);
}

struct Scatter { bool scattered; V3 color; V3 o, d; };  // RayHit (lib.rs:300-306)

// Bits of rt_material.reserved[0] that the library sets on its own copy of the materials (make_shade_records):
//   RT_MAT_SKIP_UV       no texture this material can reach reads the surface coordinates (u, v), so shading need not
//                        compute them;
//   RT_MAT_SOLID_INLINE  the material's texture is a SolidColor, whose colour stands in `albedo` (textured materials
//                        have no albedo of their own), so shading need not fetch the texture.
#define RT_MAT_SKIP_UV 1u
#define RT_MAT_SOLID_INLINE 2u
// The primitive and its material in one 64-byte record: shading reads both with one fetch instead of one after the other.
struct ShadeRec { rt_prim prim; rt_material mat; };
static_assert(sizeof(ShadeRec) == 64, "shading record size");
// (host side) the records of a scene description
inline void make_shade_records(const rt_scene_desc* D, std::vector<ShadeRec>& out) {
    std::vector<rt_material> mats(D->materials, D->materials + D->n_materials);
    for (auto& M : mats) {
        bool reads_uv = false;
        M.reserved[0] &= ~(RT_MAT_SKIP_UV | RT_MAT_SOLID_INLINE);
        if (M.kind == RT_MAT_LAMBERTIAN || M.kind == RT_MAT_DIFFUSE_LIGHT || M.kind == RT_MAT_ISOTROPIC) {
            const rt_texture& T0 = D->textures[M.texture];
            if (T0.kind == RT_TEX_SOLID) { M.reserved[0] |= RT_MAT_SOLID_INLINE; for (int c = 0; c < 3; ++c) M.albedo[c] = T0.color[c]; }
            std::vector<uint32_t> todo{M.texture};  // image textures are the only kind that reads (u, v)
            for (uint32_t depth = 0; depth <= RT_MAX_TEXTURE_NESTING && !todo.empty() && !reads_uv; ++depth) {
                std::vector<uint32_t> next;
                for (uint32_t t : todo) {
                    const rt_texture& T = D->textures[t];
                    if (T.kind == RT_TEX_IMAGE) reads_uv = true;
                    else if (T.kind == RT_TEX_CHECKERED) { next.push_back(T.odd); next.push_back(T.even); }
                }
                todo.swap(next);
            }
        }
        if (!reads_uv) M.reserved[0] |= RT_MAT_SKIP_UV;
    }
    out.resize(D->n_prims);
    for (uint32_t i = 0; i < D->n_prims; ++i) { out[i].prim = D->prims[i]; out[i].mat = mats[D->prims[i].material]; }
}
RT_HD V3 material_color(const SceneView& S, const rt_material& M, const Hit& h) {
    if (M.reserved[0] & RT_MAT_SOLID_INLINE) return v3(M.albedo[0], M.albedo[1], M.albedo[2]);
    return texture_color(S, M.texture, h.u, h.v, h.point);
}

// Material::scatter (materials.rs:28-114) with the draws of stream 0 of the segment.
RT_HD Scatter material_scatter(const SceneView& S, const rt_material& M, const Hit& h_in, const DrawKey& key) {
    const U4 rnd = philox4x32_10(key.pixel, key.sample, key.bounce, 0u, key.k0, key.k1);
    Hit h = h_in;
    bool front = true;  // IntersectionPoint::face (shape.rs:552-558); only the dielectric asks
    if (M.kind == RT_MAT_DIELECTRIC) front = dot(h.in_vec, h.normal) < 0.0f;
    if (M.reserved[1] & RT_MATF_FACE_NORMAL) {  // (not the reference: the normal turned against the incoming ray; in_vec may be unnormalised here, its sign is what counts)
        if (!(dot(h.in_vec, h.normal) < 0.0f)) h.normal = -h.normal;
    }
    Scatter out;
    out.o = h.point;
    out.d = v3(0.0f, 0.0f, 0.0f);
    if (M.kind == RT_MAT_LAMBERTIAN) {
        // direction = random_unit_vector + normal; attenuation = tex * max(cos/pi, 0) * pi / cos with
        // cos = normal . normed(direction) (materials.rs:31-44).  The sum cancels when the random vector is
        // nearly opposite to the normal, so it is formed in f64 like the reference's.
        D3 nd = d3(h.normal);
        D3 dir = random_unit_vector_d(u01(rnd.x), u01(rnd.y)) + nd;
        // weight = max(cos / pi, 0) * pi / cos, in the reference's f64 and rounded to fp32: 1 for cos > 0 (the quotient is
        // within an f64 ulp of 1), 0 for cos < 0, NaN for cos = 0; the sign of cos is the sign of normal . direction
        double proj = dot(nd, dir);
        float w = proj > 0.0 ? 1.0f : (proj < 0.0 ? -0.0f : RT_NAN_F);
        V3 tex = material_color(S, M, h);
        out.color = tex * w;
        out.scattered = true; out.d = v3(dir);
    } else if (M.kind == RT_MAT_METAL) {
        V3 refl = reflect(h.in_vec, h.normal);
        V3 s = refl + random_unit_vector(u01(rnd.x), u01(rnd.y)) * M.param;  // fuzz is not clamped
        if (dot(s, h.normal) >= 0.0f) { out.scattered = true; out.color = v3(M.albedo[0], M.albedo[1], M.albedo[2]); out.d = s; }
        else { out.scattered = false; out.color = v3(0.0f, 0.0f, 0.0f); }
    } else if (M.kind == RT_MAT_DIELECTRIC) {
        float eta = front ? 1.0f / M.param : M.param;
        V3 in_unit = normed(h.in_vec);
        float cos_theta = -fminf(dot(in_unit, h.normal), 1.0f);
        float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
        V3 dir;
        if (eta * sin_theta > 1.0f || u01(rnd.z) < schlick(cos_theta, eta)) dir = reflect(in_unit, h.normal);
        else dir = refract(in_unit, h.normal, eta);
        out.scattered = true; out.color = v3(M.albedo[0], M.albedo[1], M.albedo[2]); out.d = dir;
    } else if (M.kind == RT_MAT_DIFFUSE_LIGHT) {
        out.scattered = false; out.color = material_color(S, M, h);
    } else {  // Isotropic
        out.scattered = true; out.d = random_unit_vector(u01(rnd.x), u01(rnd.y));
        out.color = material_color(S, M, h);
    }
    return out;
}
RT_HD Scatter material_scatter(const SceneView& S, const Hit& h, const DrawKey& key) {
    return material_scatter(S, S.shade[h.prim].mat, h, key);
}


// One step of ray_color (lib.rs:89-101) for a ray whose closest hit is known.
// Returns true when the path continues (r and thr updated); otherwise `value` is the
// sample's final radiance.
RT_HD bool shade_step(const SceneView& S, Ray& r, V3& thr, const DrawKey& key, bool hit, float t, uint32_t prim, int side, bool last_bounce, V3& value) {
    if (!hit) { value = thr * v3(S.background[0], S.background[1], S.background[2]); return false; }  // lib.rs:91
    Hit h;
    const ShadeRec R = S.shade[prim];
    const rt_prim P = R.prim;
    const rt_material M = R.mat;
    prim_surface(S, P, prim, r.o, r.d, r.time, t, side, !(M.reserved[0] & RT_MAT_SKIP_UV), M.kind == RT_MAT_METAL || M.kind == RT_MAT_DIELECTRIC, h);
    Scatter sc = material_scatter(S, M, h, key);
    if (!sc.scattered) { value = thr * sc.color; return false; }  // lib.rs:92
    thr = thr * sc.color;
    r.o = sc.o; r.d = sc.d;
    if (RT_ANY_LANE(h.point_side != 0u)) if (h.point_side != 0u) r.o = side_of_surface(sc.o, h.point_side, h.normal, !(dot(sc.d, h.normal) < 0.0f));
    // lib.rs:103: a path that is still alive after max_reflections segments is black.  A path whose
    // throughput is exactly zero can only ever contribute zero, so it ends here as well.
    if (last_bounce || (thr.x == 0.0f && thr.y == 0.0f && thr.z == 0.0f)) { value = v3(0.0f, 0.0f, 0.0f); return false; }
    return true;
}

// ------------------------------------------------------------------ camera
// pixel() / Camera::ray / random_time (lib.rs:55-60, :389-401) in f64, rounded
// to fp32 at the end.  (i, j) = (column, row counted from the bottom).
RT_HD Ray camera_ray(const rt_camera& C, uint32_t i, uint32_t j, uint32_t width, uint32_t height, uint32_t sample, uint32_t k0, uint32_t k1) {
    const uint32_t pixel = j * width + i;
    const U4 r0 = philox4x32_10(pixel, sample, RT_TAG_CAMERA, 0u, k0, k1);
    const double s = ((double)i + (double)u01(r0.x)) / (double)(width - 1);
    const double t = ((double)j + (double)u01(r0.y)) / (double)(height - 1);
    double dx, dy;
    for (uint32_t round = 1;; ++round) {  // InsideUnitDisc (geometry.rs:397-408), two candidates per Philox block
        const U4 q = philox4x32_10(pixel, sample, RT_TAG_CAMERA, round, k0, k1);
        // (norm >= 1 decided on the squared norm: the candidates are odd multiples of 2^-23, so their squares and the sum
        // are exact multiples of 2^-46, and such a sum is below 1 iff its correctly rounded square root is)
        dx = -1.0 + 2.0 * (double)u01(q.x); dy = -1.0 + 2.0 * (double)u01(q.y);
        if (!(dx * dx + dy * dy >= 1.0)) break;
        dx = -1.0 + 2.0 * (double)u01(q.z); dy = -1.0 + 2.0 * (double)u01(q.w);
        if (!(dx * dx + dy * dy >= 1.0)) break;
    }
    dx *= C.lens_radius; dy *= C.lens_radius;
    D3 offset = d3(C.u) * dx + d3(C.v) * dy;
    D3 org = d3(C.origin) + offset;
    D3 dir = (d3(C.lower_left_corner) + d3(C.horizontal) * s + d3(C.vertical) * t) - org;
    Ray r;
    r.o = v3(org); r.d = v3(dir);
    r.time = (float)(C.shutter_open + (C.shutter_close - C.shutter_open) * (double)u01(r0.z));
    return r;
}

// lib.rs:68-72: x.max(0).min(1).sqrt() * 255.999 as u8 (saturating, NaN -> 0).
RT_HD uint8_t quantise(double x) {
    double y = sqrt(fmin(fmax(x, 0.0), 1.0)) * 255.999;
    if (!(y > 0.0)) return 0;
    if (y >= 255.0) return 255;
    return (uint8_t)y;
}

// ----------------------------------------------------------- tile ownership
// The frame is cut into tiles of tw x th pixels, numbered tile_id = ty * ntx + tx over (column, row counted from the bottom);
// rank `rank` of `world` renders the tiles with tile_id % world == rank.  Edge tiles are cut off at the frame's border.
struct TileGrid {
    uint32_t W, H, tw, th, rank, world;
    RT_HD uint32_t ntx() const { return (W + tw - 1) / tw; }
    RT_HD uint32_t nty() const { return (H + th - 1) / th; }
    RT_HD uint64_t ntiles() const { return (uint64_t)ntx() * nty(); }
};
// Numbers t in [0, x) with t % m == r.
RT_HD uint64_t count_residue_below(uint64_t x, uint32_t r, uint32_t m) { return x > r ? (x - r - 1) / m + 1 : 0; }
// Whether the last tile of tile row `ty` belongs to the rank and has a number below t_end (the one term of the count below
// that has no closed form: callers sum it over the tile rows, the device spread over a workgroup's threads).
RT_HD uint32_t owned_last_column_tile(const TileGrid& G, uint32_t ty, uint64_t t_end) {
    const uint64_t t = (uint64_t)ty * G.ntx() + G.ntx() - 1;
    return (t < t_end && t % G.world == G.rank) ? 1u : 0u;
}
// Pixels in the rank's tiles with a number below t_end (t_end = ntiles(): all of the rank's pixels), given
// last_column = the sum of owned_last_column_tile over all tile rows.  Every owned tile counts tw * th, less the cut of the
// last column and of the last row.
RT_HD uint64_t owned_pixels_below(const TileGrid& G, uint64_t t_end, uint64_t last_column) {
    const uint64_t ntx = G.ntx(), nty = G.nty(), n = ntx * nty;
    if (t_end > n) t_end = n;
    const uint64_t dw = ntx * G.tw - G.W, dh = nty * G.th - G.H;  // what the last column / row lack of a full tile
    const uint64_t all = count_residue_below(t_end, G.rank, G.world);
    const uint64_t row0 = (nty - 1) * ntx;  // first tile of the last row
    const uint64_t last_row = count_residue_below(t_end, G.rank, G.world) - count_residue_below(row0 < t_end ? row0 : t_end, G.rank, G.world);
    const uint64_t corner = (n - 1 < t_end && (n - 1) % G.world == G.rank) ? 1 : 0;
    return all * G.tw * G.th - last_column * dw * G.th - last_row * G.tw * dh + corner * dw * dh;
}
inline uint64_t owned_pixels_below(const TileGrid& G, uint64_t t_end) {  // (host side)
    uint64_t last_column = 0;
    for (uint32_t ty = 0; ty < G.nty(); ++ty) last_column += owned_last_column_tile(G, ty, t_end);
    return owned_pixels_below(G, t_end, last_column);
}

}  // namespace rt
#endif
