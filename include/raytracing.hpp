// raytracing.hpp -- C++ host API mirroring the public API of the reference crate
// (type, variant and field names follow the Rust ones), so that a scene script
// reads like examples/*.rs.  The host side builds the scene tree, flattens it
// (rt_scene_desc of rt_abi.h) and hands it to the HIP library; it does no
// rendering arithmetic of its own.
//
//   Rust (reference)                                  C++ (this header)
//   geometry::{Point3, Vec3, Axis, Shape}             raytracing::geometry::...
//   Shape::Sphere { center, radius }                  Shape::Sphere(center, radius)
//   Shape::sphere(r, (t0, c0), (t1, c1))              Shape::sphere(r, {t0, c0}, {t1, c1})
//   Shape::rectangle / new_box                        same
//   .flipped() .translated(v) .rotated(axis, deg)     same
//   Shape::ConstantMedium { boundary, density }       Shape::ConstantMedium(boundary, density)
//   textures::Texture::{SolidColor, Checkered, Noise, Image}, Texture::image(path)
//   materials::Material::{Lambertian, Metal, Dielectric, DiffuseLight, Isotropic}, Material::glass()
//   Object { shape, material }, Scene::new(objects, background, &mut rng)
//   camera::{Camera::new(CameraOptions), CameraOptions}, ImageOptions, pixels(&camera, &scene, opts)
//
// Where the reference panics (empty scene, Shape::sphere with end_time <=
// start_time, unreadable texture file, empty shutter interval) these functions
// throw std::invalid_argument / std::runtime_error.
#ifndef RAYTRACING_HPP
#define RAYTRACING_HPP

#include <array>
#include <cstdint>
#include <memory>
#include <optional>
#include <string>
#include <utility>
#include <vector>

#include "rt_abi.h"

namespace raytracing {

// Seeded generator standing in for the `&mut R: Rng` arguments of the reference
// (which passes the OS-seeded thread_rng): SplitMix64, uniform f64 from the top 53 bits.
class Rng {
public:
    explicit Rng(uint64_t seed) : s_(seed) {}
    uint64_t next_u64();
    double gen();                             // rng.gen::<f64>()  in [0,1)
    double gen_range(double lo, double hi);   // rng.gen_range(lo, hi)
private:
    uint64_t s_;
};

namespace geometry {

enum class Axis { X = 0, Y = 1, Z = 2 };

struct Vec3 {  // src/geometry.rs:74
    std::array<double, 3> v{0, 0, 0};
    Vec3() = default;
    Vec3(double x, double y, double z) : v{x, y, z} {}
    double& operator[](size_t i) { return v[i]; }
    double operator[](size_t i) const { return v[i]; }
    double norm() const;
    double norm_squared() const;
    double dot(const Vec3& o) const;
    Vec3 cross(const Vec3& o) const;
    Vec3 normed() const;
    static Vec3 random(double lo, double hi, Rng& rng);  // Vec3::random(lo..hi, &mut rng), geometry.rs:106
};
Vec3 operator+(const Vec3& a, const Vec3& b);
Vec3 operator-(const Vec3& a, const Vec3& b);
Vec3 operator-(const Vec3& a);
Vec3 operator*(const Vec3& a, double s);
Vec3 operator*(const Vec3& a, const Vec3& b);
Vec3 operator/(const Vec3& a, double s);

struct Point3 {  // src/geometry.rs:25
    std::array<double, 3> v{0, 0, 0};
    Point3() = default;
    Point3(double x, double y, double z) : v{x, y, z} {}
    double& operator[](size_t i) { return v[i]; }
    double operator[](size_t i) const { return v[i]; }
    double dist(const Point3& o) const;
    bool operator==(const Point3& o) const { return v == o.v; }
};
Point3 operator+(const Point3& a, const Vec3& b);
Point3 operator-(const Point3& a, const Vec3& b);
Vec3 operator-(const Point3& a, const Point3& b);

struct BoundingBox {  // src/geometry/shape.rs:571-576
    Point3 min, max;
};
BoundingBox operator+(const BoundingBox& a, const BoundingBox& b);

// src/geometry/shape.rs:62-106.  A Shape is an immutable value; copies share the tree.
class Shape {
public:
    enum class Kind { Sphere, MovingSphere, Rectangle, ConstantMedium, Box, Flipped, Translated, Rotated };
    struct Node;

    static Shape Sphere(const Point3& center, double radius);
    // NOT in the reference: a sphere that is also hit from within, at its far root (RT_PRIM_BOTH_ROOTS in rt_abi.h); for
    // comparisons with the book the crate follows.
    static Shape SphereBothRoots(const Point3& center, double radius);
    Shape with_both_roots() const;  // this shape with every sphere in it made such a sphere (not the reference's behaviour either)
    static Shape sphere(double radius, std::pair<double, Point3> start, std::pair<double, Point3> end);  // shape.rs:222
    static Shape rectangle(const Point3& lower_left, const Point3& upper_right);                         // shape.rs:129
    static Shape new_box(const Point3& min, const Point3& max);                                          // shape.rs:157
    static Shape ConstantMedium(const Shape& boundary, double density);
    // NOT in the reference: a medium whose boundary queries see the ray behind its origin where it is (RT_PRIM_MULTI_SCATTER in
    // rt_abi.h), so that a path scatters inside it again and again; the reference's media scatter at most once per entry.
    static Shape ConstantMediumMultiScatter(const Shape& boundary, double density);
    Shape flipped() const;                          // shape.rs:110
    Shape translated(const Vec3& offset) const;     // shape.rs:115
    Shape rotated(Axis axis, double angle) const;   // shape.rs:123 (degrees)
    Shape rotate(Axis axis, double angle) const { return rotated(axis, angle); }

    std::optional<BoundingBox> bound() const;  // Boundable::bound, shape.rs:428-524
    Kind kind() const;
    const Node& node() const { return *n_; }

private:
    explicit Shape(std::shared_ptr<const Node> n) : n_(std::move(n)) {}
    std::shared_ptr<const Node> n_;
};

}  // namespace geometry

namespace noise {
// Stand-in for `noise::Perlin` of the noise 0.6.0 crate: the permutation table
// is explicit so that a binding can inject the crate's real table.
struct Perlin {
    std::array<uint8_t, 256> table;
    Perlin();                                  // Perlin::new(): seed 0
    explicit Perlin(uint32_t seed);
    explicit Perlin(const std::array<uint8_t, 256>& t) : table(t) {}
};
// Stand-ins for the crate's `Fbm` and `Turbulence<Perlin>` with the crate's defaults (Fbm: 6 octaves, lacunarity 2 pi / 3,
// persistence 0.5, octave i seeded with seed + i; Turbulence: power 1, roughness 3, distortion functions seeded with seed,
// seed + 1, seed + 2; frequencies are 1).  No example of the reference uses them: `Texture::Noise` takes any `NoiseFn`
// (textures.rs:14-17), these are the two more the device can evaluate.
struct Fbm {
    uint32_t seed = 0, octaves = 6;
    double lacunarity = 2.0943951023931953, persistence = 0.5;
};
struct Turbulence {
    Perlin source;
    uint32_t seed = 0, roughness = 3;
    double power = 1.0, lacunarity = 2.0943951023931953, persistence = 0.5;  // (lacunarity, persistence: of the distortion functions)
};
}  // namespace noise

namespace textures {
using geometry::Vec3;
class Texture {  // src/textures.rs:8-19
public:
    enum class Kind { SolidColor, Checkered, Noise, Image };
    struct Node;
    static Texture SolidColor(const Vec3& color);
    static Texture Checkered(const Texture& odd, const Texture& even);
    static Texture Noise(double scale, const noise::Perlin& noise);
    static Texture Noise(double scale, const noise::Fbm& noise);
    static Texture Noise(double scale, const noise::Turbulence& noise);
    static Texture Image(uint32_t width, uint32_t height, std::vector<uint8_t> rgb);
    static Texture image(const std::string& path);  // textures.rs:41; reads binary PPM (P6), 8 bit
    const Node& node() const { return *n_; }
private:
    explicit Texture(std::shared_ptr<const Node> n) : n_(std::move(n)) {}
    std::shared_ptr<const Node> n_;
};
}  // namespace textures

namespace materials {
using geometry::Vec3;
using textures::Texture;
class Material {  // src/materials.rs:8-18
public:
    enum class Kind { Lambertian, Metal, Dielectric, DiffuseLight, Isotropic };
    static Material Lambertian(const Texture& texture);
    static Material Metal(const Vec3& albedo, double fuzz);
    static Material Dielectric(const Vec3& albedo, double refraction_index);
    static Material DiffuseLight(const Texture& emit);
    static Material Isotropic(const Texture& texture);
    static Material glass();  // materials.rs:21-26
    // NOT in the reference: this material, handed the normal turned against the incoming ray (RT_MATF_FACE_NORMAL in rt_abi.h).
    Material facing_normal() const { Material m = *this; m.face_normal = true; return m; }
    bool face_normal = false;
    Kind kind;
    std::optional<Texture> texture;
    Vec3 albedo;
    double param = 0.0;  // fuzz | refraction_index
private:
    Material() : kind(Kind::Lambertian) {}
};
}  // namespace materials

struct Object {  // src/lib.rs:261-266
    geometry::Shape shape;
    materials::Material material;
};

// Options of the host-side flattener (no counterpart in the reference).
struct FlattenOptions {
    bool keep_inert = false;  // keep provably-unhittable primitives (sphere-bounded media) in the BVH
};

// src/lib.rs:119-133.  Construction flattens the object list (primitive id = index in `objects`)
// and builds the BVH; the device copy is created on first use by pixels().
class Scene {
public:
    Scene(std::vector<Object> objects, const geometry::Vec3& background_color, Rng& rng, const FlattenOptions& opt = FlattenOptions());
    ~Scene();
    Scene(const Scene&) = delete;
    Scene& operator=(const Scene&) = delete;
    const rt_scene_desc& desc() const;
    rt_scene* device_scene(int device = -1) const;  // throws std::runtime_error when the HIP library reports an error
    size_t num_objects() const;
    struct Impl;
private:
    std::unique_ptr<Impl> impl_;
};

namespace camera {
using geometry::Point3;
using geometry::Vec3;
struct CameraOptions {  // src/lib.rs:405-425
    double aperture, aspect_ratio, focus_distance;
    Point3 looking_at, origin;
    double shutter_close, shutter_open, vfov;
    Vec3 vup;
};
class Camera {  // src/lib.rs:314-384
public:
    explicit Camera(const CameraOptions& options);
    const rt_camera& abi() const { return cam_; }
private:
    rt_camera cam_;
};
}  // namespace camera

struct ImageOptions {  // src/lib.rs:108-117
    uint32_t height, width;
    size_t samples_per_pixel, max_reflections;
};

// Seed / device selection for pixels(); the reference has neither.
struct RenderOptions {
    uint64_t seed = 1;
    int device = -1;
    // More than one entry: the frame's 16x16 tiles are dealt round robin to these devices (one host thread and one
    // device copy of the scene each), the per-device frames of f64 means are added on the host (adding zeros is
    // exact) and quantised there; the bytes are those of a single-device render.
    std::vector<int> devices;
};

// src/lib.rs:18 -- 3 bytes per pixel, row-major, top row first.  Runs on the GPU
// through the C ABI; throws std::runtime_error if the library reports an error.
std::vector<uint8_t> pixels(const camera::Camera& camera, const Scene& scene, ImageOptions image_options, const RenderOptions& render_options = RenderOptions());

// The five scene definitions of examples/*.rs (random parameters drawn from `rng` in source order).
namespace examples {
struct Example {
    std::vector<Object> objects;
    geometry::Vec3 background;
    camera::CameraOptions camera;
};
Example random_spheres(Rng& rng);
Example cornell();
Example perlin();
Example cornell_smoke();
Example final_scene(Rng& rng, const textures::Texture& earthmap);
Example final_scene_book(Rng& rng, const textures::Texture& earthmap);  // NOT a reference example: final_scene from the book's camera, far-root spheres, facing normals
Example by_name(const std::string& name, Rng& rng, const textures::Texture* earthmap);  // the five names above, or "final_scene_book"
}  // namespace examples

// lib.rs:68-72: x.max(0).min(1).sqrt() * 255.999 as u8 (saturating, NaN -> 0).
uint8_t quantise(double mean);

// Writes an 8-bit RGB image as binary PPM / as PNG (stored deflate blocks, no compression); they stand in for
// ImageBuffer::from_raw(..).save("test.png") of the examples (e.g. cornell.rs:109-116).
void write_ppm(const std::string& path, uint32_t width, uint32_t height, const std::vector<uint8_t>& rgb);
void write_png(const std::string& path, uint32_t width, uint32_t height, const std::vector<uint8_t>& rgb);

}  // namespace raytracing
#endif
