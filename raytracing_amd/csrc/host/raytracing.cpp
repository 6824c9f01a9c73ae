// raytracing.cpp -- host side of the renderer: the scene-construction API of
// include/raytracing.hpp, the flattener (scene tree -> rt_scene_desc), the BVH
// builder, Camera::new and the pixels() entry that drives the HIP library.
// No intersection or shading arithmetic lives here.

#include "raytracing.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <thread>
#include <limits>
#include <stdexcept>

namespace raytracing {

static const double PI = 3.14159265358979323846264338327950288;

// ---------------------------------------------------------------------- Rng
uint64_t Rng::next_u64() {
    uint64_t z = (s_ += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
double Rng::gen() { return (double)(next_u64() >> 11) * (1.0 / 9007199254740992.0); }
double Rng::gen_range(double lo, double hi) { return lo + (hi - lo) * gen(); }

// ----------------------------------------------------------------- geometry
namespace geometry {

double Vec3::norm_squared() const { return v[0] * v[0] + v[1] * v[1] + v[2] * v[2]; }
double Vec3::norm() const { return std::sqrt(norm_squared()); }
double Vec3::dot(const Vec3& o) const { return v[0] * o[0] + v[1] * o[1] + v[2] * o[2]; }
Vec3 Vec3::cross(const Vec3& o) const { return Vec3(v[1] * o[2] - v[2] * o[1], v[2] * o[0] - v[0] * o[2], v[0] * o[1] - v[1] * o[0]); }
Vec3 Vec3::normed() const { return *this / norm(); }
Vec3 Vec3::random(double lo, double hi, Rng& rng) {
    double a = rng.gen_range(lo, hi), b = rng.gen_range(lo, hi), c = rng.gen_range(lo, hi);
    return Vec3(a, b, c);
}
Vec3 operator+(const Vec3& a, const Vec3& b) { return Vec3(a[0] + b[0], a[1] + b[1], a[2] + b[2]); }
Vec3 operator-(const Vec3& a, const Vec3& b) { return Vec3(a[0] - b[0], a[1] - b[1], a[2] - b[2]); }
Vec3 operator-(const Vec3& a) { return Vec3(-a[0], -a[1], -a[2]); }
Vec3 operator*(const Vec3& a, double s) { return Vec3(a[0] * s, a[1] * s, a[2] * s); }
Vec3 operator*(const Vec3& a, const Vec3& b) { return Vec3(a[0] * b[0], a[1] * b[1], a[2] * b[2]); }
Vec3 operator/(const Vec3& a, double s) { return Vec3(a[0] / s, a[1] / s, a[2] / s); }
double Point3::dist(const Point3& o) const { return (*this - o).norm(); }
Point3 operator+(const Point3& a, const Vec3& b) { return Point3(a[0] + b[0], a[1] + b[1], a[2] + b[2]); }
Point3 operator-(const Point3& a, const Vec3& b) { return Point3(a[0] - b[0], a[1] - b[1], a[2] - b[2]); }
Vec3 operator-(const Point3& a, const Point3& b) { return Vec3(a[0] - b[0], a[1] - b[1], a[2] - b[2]); }
BoundingBox operator+(const BoundingBox& a, const BoundingBox& b) {
    BoundingBox r;
    for (int c = 0; c < 3; ++c) { r.min[c] = std::fmin(a.min[c], b.min[c]); r.max[c] = std::fmax(a.max[c], b.max[c]); }
    return r;
}

struct Shape::Node {
    Kind kind;
    Point3 center; double radius = 0;                 // Sphere; MovingSphere: center = ray.origin
    bool both_roots = false;                          // Sphere: the far-root variant (not the reference's)
    bool multi_scatter = false;                       // ConstantMedium: boundary queries without Ray::at's clamp (not the reference's)
    Vec3 motion; double start_time = 0, end_time = 0; // MovingSphere: ray.direction
    Axis axis = Axis::X; double x0 = 0, y0 = 0, x1 = 0, y1 = 0, height = 0;  // Rectangle; Rotated uses axis
    Point3 bmin, bmax;                                // Box
    std::shared_ptr<const Node> inner;                // wrappers / medium boundary
    Vec3 offset; double angle = 0; double density = 0;
};

static void axis_perm(Axis axis, int& p1, int& p2, int& o) {
    switch (axis) {
        case Axis::X: p1 = 1; p2 = 2; o = 0; break;
        case Axis::Y: p1 = 2; p2 = 0; o = 1; break;
        default: p1 = 0; p2 = 1; o = 2; break;
    }
}

Shape Shape::Sphere(const Point3& center, double radius) {
    auto n = std::make_shared<Node>(); n->kind = Kind::Sphere; n->center = center; n->radius = radius;
    return Shape(n);
}
Shape Shape::SphereBothRoots(const Point3& center, double radius) {
    auto n = std::make_shared<Node>(); n->kind = Kind::Sphere; n->center = center; n->radius = radius; n->both_roots = true;
    return Shape(n);
}
static std::shared_ptr<const Shape::Node> both_roots_copy(const std::shared_ptr<const Shape::Node>& n) {
    auto c = std::make_shared<Shape::Node>(*n);
    if (c->kind == Shape::Kind::Sphere || c->kind == Shape::Kind::MovingSphere) c->both_roots = true;
    if (c->inner) c->inner = both_roots_copy(c->inner);
    return c;
}
Shape Shape::with_both_roots() const { return Shape(both_roots_copy(n_)); }
Shape Shape::sphere(double radius, std::pair<double, Point3> start, std::pair<double, Point3> end) {
    if (!(end.first > start.first)) throw std::invalid_argument("Shape::sphere: end_time must be greater than start_time");
    if (start.second == end.second) return Sphere(start.second, radius);
    auto n = std::make_shared<Node>();
    n->kind = Kind::MovingSphere; n->center = start.second; n->motion = end.second - start.second;
    n->radius = radius; n->start_time = start.first; n->end_time = end.first;
    return Shape(n);
}
Shape Shape::rectangle(const Point3& ll, const Point3& ur) {
    Axis axis = ll[0] == ur[0] ? Axis::X : (ll[1] == ur[1] ? Axis::Y : Axis::Z);
    int p1, p2, o; axis_perm(axis, p1, p2, o);
    auto n = std::make_shared<Node>();
    n->kind = Kind::Rectangle; n->axis = axis; n->height = ll[o];
    n->x0 = ll[p1]; n->y0 = ll[p2]; n->x1 = ur[p1]; n->y1 = ur[p2];
    return Shape(n);
}
Shape Shape::new_box(const Point3& mn, const Point3& mx) {
    auto n = std::make_shared<Node>(); n->kind = Kind::Box; n->bmin = mn; n->bmax = mx;
    return Shape(n);
}
Shape Shape::ConstantMedium(const Shape& boundary, double density) {
    auto n = std::make_shared<Node>(); n->kind = Kind::ConstantMedium; n->inner = boundary.n_; n->density = density;
    return Shape(n);
}
Shape Shape::ConstantMediumMultiScatter(const Shape& boundary, double density) {
    auto n = std::make_shared<Node>(); n->kind = Kind::ConstantMedium; n->inner = boundary.n_; n->density = density; n->multi_scatter = true;
    return Shape(n);
}
Shape Shape::flipped() const {
    auto n = std::make_shared<Node>(); n->kind = Kind::Flipped; n->inner = n_;
    return Shape(n);
}
Shape Shape::translated(const Vec3& offset) const {
    auto n = std::make_shared<Node>(); n->kind = Kind::Translated; n->inner = n_; n->offset = offset;
    return Shape(n);
}
Shape Shape::rotated(Axis axis, double angle) const {
    auto n = std::make_shared<Node>(); n->kind = Kind::Rotated; n->inner = n_; n->axis = axis; n->angle = angle;
    return Shape(n);
}
Shape::Kind Shape::kind() const { return n_->kind; }

static Vec3 rotate_deg(const Vec3& s, Axis axis, double angle_deg) {  // geometry.rs:112-129
    double a = angle_deg * (PI / 180.0), c = std::cos(a), sn = std::sin(a);
    int c0, c1;
    switch (axis) { case Axis::X: c0 = 1; c1 = 2; break; case Axis::Y: c0 = 2; c1 = 0; break; default: c0 = 0; c1 = 1; break; }
    Vec3 out = s;
    out[c0] = s[c0] * c - s[c1] * sn;
    out[c1] = s[c0] * sn + s[c1] * c;
    return out;
}

static BoundingBox bound_of(const Shape::Node& n) {  // shape.rs:428-524 (needed to build the BVH)
    using K = Shape::Kind;
    switch (n.kind) {
        case K::Sphere: {
            BoundingBox b;
            for (int c = 0; c < 3; ++c) { b.min[c] = n.center[c] - 1.0 * n.radius; b.max[c] = n.center[c] + 1.0 * n.radius; }
            return b;
        }
        case K::MovingSphere: {
            Shape::Node a; a.kind = K::Sphere; a.center = n.center; a.radius = n.radius;
            Shape::Node b = a;
            double dt = n.end_time - n.start_time;  // ray.at(dt): the clamp of Ray::at cannot trigger (dt > 0)
            b.center = n.center + n.motion * dt;
            return bound_of(a) + bound_of(b);
        }
        case K::Rectangle: {
            int p1, p2, o; axis_perm(n.axis, p1, p2, o);
            BoundingBox b;
            b.min[p1] = n.x0; b.min[p2] = n.y0; b.min[o] = n.height - 0.0001;
            b.max[p1] = n.x1; b.max[p2] = n.y1; b.max[o] = n.height + 0.0001;
            return b;
        }
        case K::Box: return BoundingBox{n.bmin, n.bmax};
        case K::Flipped: case K::ConstantMedium: return bound_of(*n.inner);
        case K::Translated: { BoundingBox b = bound_of(*n.inner); return BoundingBox{b.min + n.offset, b.max + n.offset}; }
        case K::Rotated: {
            BoundingBox bb = bound_of(*n.inner), r;
            const double inf = std::numeric_limits<double>::infinity();
            r.min = Point3(inf, inf, inf); r.max = Point3(-inf, -inf, -inf);
            for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int k = 0; k < 2; ++k) {
                Vec3 t = rotate_deg(Vec3(i ? bb.max[0] : bb.min[0], j ? bb.max[1] : bb.min[1], k ? bb.max[2] : bb.min[2]), n.axis, n.angle);
                for (int c = 0; c < 3; ++c) { r.min[c] = std::fmin(r.min[c], t[c]); r.max[c] = std::fmax(r.max[c], t[c]); }
            }
            return r;
        }
    }
    return BoundingBox{};
}
std::optional<BoundingBox> Shape::bound() const { return bound_of(*n_); }

}  // namespace geometry

// -------------------------------------------------------------------- noise
namespace noise {
// Permutation table of noise 0.6.0's PermutationTable::new(seed): 0..=255 shuffled (rand 0.7
// SliceRandom::shuffle) by an XorShiftRng seeded with the bytes [1,0,0,0, seed, seed, seed].
// Restated from the crates' published sources from memory; not verifiable offline.
std::array<uint8_t, 256> table_from_seed(uint32_t seed) {
    uint32_t s[4] = {1u, seed, seed, seed};
    auto next = [&s]() {
        uint32_t x = s[0], t = x ^ (x << 11);
        s[0] = s[1]; s[1] = s[2]; s[2] = s[3];
        uint32_t w = s[3];
        s[3] = w ^ (w >> 19) ^ (t ^ (t >> 8));
        return s[3];
    };
    std::array<uint8_t, 256> tbl;
    for (int i = 0; i < 256; ++i) tbl[i] = (uint8_t)i;
    for (uint32_t i = 255; i >= 1; --i) {
        uint32_t range = i + 1, zone = (range << __builtin_clz(range)) - 1u, idx;
        for (;;) {
            uint64_t m = (uint64_t)next() * range;
            if ((uint32_t)m <= zone) { idx = (uint32_t)(m >> 32); break; }
        }
        std::swap(tbl[i], tbl[idx]);
    }
    return tbl;
}
Perlin::Perlin() : table(table_from_seed(0)) {}
Perlin::Perlin(uint32_t seed) : table(table_from_seed(seed)) {}
}  // namespace noise

// ----------------------------------------------------------------- textures
namespace textures {
struct Texture::Node {
    Kind kind;
    Vec3 color;
    std::shared_ptr<const Node> odd, even;
    double scale = 1.0; std::array<uint8_t, 256> table{};  // table: Perlin's, or a Turbulence's source
    uint32_t function = RT_NOISE_PERLIN, octaves = 0; double lacunarity = 0.0, persistence = 0.0, power = 0.0;
    std::vector<uint8_t> octave_tables;                    // Fbm / Turbulence: the consecutive tables of seed, seed + 1, ...
    uint32_t width = 0, height = 0; std::vector<uint8_t> rgb;
};
Texture Texture::SolidColor(const Vec3& color) { auto n = std::make_shared<Node>(); n->kind = Kind::SolidColor; n->color = color; return Texture(n); }
Texture Texture::Checkered(const Texture& odd, const Texture& even) {
    auto n = std::make_shared<Node>(); n->kind = Kind::Checkered; n->odd = odd.n_; n->even = even.n_; return Texture(n);
}
Texture Texture::Noise(double scale, const noise::Perlin& p) {
    auto n = std::make_shared<Node>(); n->kind = Kind::Noise; n->scale = scale; n->table = p.table; return Texture(n);
}
static std::vector<uint8_t> tables_from_seeds(uint32_t seed, uint32_t count) {
    std::vector<uint8_t> out;
    for (uint32_t i = 0; i < count; ++i) { const auto t = noise::table_from_seed(seed + i); out.insert(out.end(), t.begin(), t.end()); }
    return out;
}
Texture Texture::Noise(double scale, const noise::Fbm& f) {
    if (f.octaves < 1 || f.octaves > 32) throw std::invalid_argument("Texture::Noise: an Fbm has 1 to 32 octaves");
    auto n = std::make_shared<Node>(); n->kind = Kind::Noise; n->scale = scale; n->function = RT_NOISE_FBM; n->octaves = f.octaves;
    n->lacunarity = f.lacunarity; n->persistence = f.persistence; n->octave_tables = tables_from_seeds(f.seed, f.octaves);
    return Texture(n);
}
Texture Texture::Noise(double scale, const noise::Turbulence& t) {
    if (t.roughness < 1 || t.roughness > 32) throw std::invalid_argument("Texture::Noise: a Turbulence has roughness 1 to 32");
    auto n = std::make_shared<Node>(); n->kind = Kind::Noise; n->scale = scale; n->function = RT_NOISE_TURBULENCE; n->octaves = t.roughness;
    n->lacunarity = t.lacunarity; n->persistence = t.persistence; n->power = t.power; n->table = t.source.table;
    n->octave_tables = tables_from_seeds(t.seed, t.roughness + 2);  // three distortion functions, seeds seed, seed + 1, seed + 2
    return Texture(n);
}
Texture Texture::Image(uint32_t width, uint32_t height, std::vector<uint8_t> rgb) {
    if (width == 0 || height == 0 || rgb.size() != (size_t)3 * width * height) throw std::invalid_argument("Texture::Image: size mismatch");
    auto n = std::make_shared<Node>(); n->kind = Kind::Image; n->width = width; n->height = height; n->rgb = std::move(rgb); return Texture(n);
}
Texture Texture::image(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("Texture::image: cannot open " + path);
    std::string magic; f >> magic;
    if (magic != "P6") throw std::runtime_error("Texture::image: only binary PPM (P6) files are supported: " + path);
    auto next_int = [&f]() {
        for (;;) {
            int c = f.peek();
            if (c == '#') { std::string line; std::getline(f, line); }
            else if (std::isspace(c)) f.get();
            else break;
        }
        long v; f >> v; return v;
    };
    long w = next_int(), h = next_int(), maxv = next_int();
    f.get();
    if (!f || w <= 0 || h <= 0 || maxv != 255) throw std::runtime_error("Texture::image: bad PPM header in " + path);
    std::vector<uint8_t> rgb((size_t)3 * w * h);
    f.read((char*)rgb.data(), (std::streamsize)rgb.size());
    if (!f) throw std::runtime_error("Texture::image: truncated PPM " + path);
    return Image((uint32_t)w, (uint32_t)h, std::move(rgb));
}
}  // namespace textures

// ---------------------------------------------------------------- materials
namespace materials {
Material Material::Lambertian(const Texture& t) { Material m; m.kind = Kind::Lambertian; m.texture = t; return m; }
Material Material::Metal(const Vec3& a, double fuzz) { Material m; m.kind = Kind::Metal; m.albedo = a; m.param = fuzz; return m; }
Material Material::Dielectric(const Vec3& a, double ri) { Material m; m.kind = Kind::Dielectric; m.albedo = a; m.param = ri; return m; }
Material Material::DiffuseLight(const Texture& t) { Material m; m.kind = Kind::DiffuseLight; m.texture = t; return m; }
Material Material::Isotropic(const Texture& t) { Material m; m.kind = Kind::Isotropic; m.texture = t; return m; }
Material Material::glass() { return Dielectric(Vec3(1.0, 1.0, 1.0), 1.5); }
}  // namespace materials

// ------------------------------------------------------------------- camera
namespace camera {
Camera::Camera(const CameraOptions& o) {  // lib.rs:347-384
    if (!(o.shutter_open < o.shutter_close)) throw std::invalid_argument("Camera: shutter_open must be below shutter_close");
    double h = std::tan((o.vfov / 2.0) * (PI / 180.0));
    double viewport_height = 2.0 * h, viewport_width = viewport_height * o.aspect_ratio;
    Vec3 w = (o.origin - o.looking_at).normed();  // Onb::from23(vup, origin - looking_at), geometry.rs:313
    Vec3 u = o.vup.cross(w).normed();
    Vec3 v = w.cross(u).normed();
    Vec3 horizontal = u * viewport_width * o.focus_distance;
    Vec3 vertical = v * viewport_height * o.focus_distance;
    Point3 llc = o.origin - horizontal / 2.0 - vertical / 2.0 - w * o.focus_distance;
    for (int c = 0; c < 3; ++c) {
        cam_.origin[c] = o.origin[c]; cam_.lower_left_corner[c] = llc[c]; cam_.horizontal[c] = horizontal[c];
        cam_.vertical[c] = vertical[c]; cam_.u[c] = u[c]; cam_.v[c] = v[c];
    }
    cam_.lens_radius = o.aperture / 2.0;
    cam_.shutter_open = o.shutter_open; cam_.shutter_close = o.shutter_close;
}
}  // namespace camera

// ---------------------------------------------------------------- flattener
using geometry::BoundingBox;
using geometry::Shape;

struct Scene::Impl {
    std::vector<rt_prim> prims;
    std::vector<rt_sphere> spheres;
    std::vector<rt_rect> rects;
    std::vector<rt_box> boxes;
    std::vector<rt_xform> xforms;
    std::vector<rt_material> materials;
    std::vector<rt_texture> textures;
    std::vector<rt_image> images;
    std::vector<uint8_t> texels;
    std::vector<uint8_t> perm_tables;
    std::vector<rt_bvh_node> bvh;
    rt_scene_desc desc{};
    mutable rt_scene* dev = nullptr;
    mutable int dev_index = -2;
    // dedup caches (node pointer -> index)
    std::vector<std::pair<const void*, uint32_t>> tex_cache, img_cache;

    uint32_t add_texture(const textures::Texture::Node& t);
    void add_object(const Object& o);
};

uint32_t Scene::Impl::add_texture(const textures::Texture::Node& t) {
    for (auto& e : tex_cache) if (e.first == &t) return e.second;
    using K = textures::Texture::Kind;
    rt_texture r{};
    switch (t.kind) {
        case K::SolidColor:
            r.kind = RT_TEX_SOLID;
            for (int c = 0; c < 3; ++c) r.color[c] = (float)t.color[c];
            break;
        case K::Checkered:
            r.kind = RT_TEX_CHECKERED; r.odd = add_texture(*t.odd); r.even = add_texture(*t.even);
            break;
        case K::Noise: {
            r.kind = RT_TEX_NOISE; r.scale = (float)t.scale;
            // a run of whole tables in perm_tables equal to `bytes`, appended if there is none
            auto place = [this](const uint8_t* bytes, size_t n_bytes) {
                for (size_t at = 0; at + n_bytes <= perm_tables.size(); at += 256)
                    if (std::memcmp(&perm_tables[at], bytes, n_bytes) == 0) return (uint32_t)(at / 256);
                const uint32_t first = (uint32_t)(perm_tables.size() / 256);
                perm_tables.insert(perm_tables.end(), bytes, bytes + n_bytes);
                return first;
            };
            if (t.function == RT_NOISE_PERLIN) r.resource = place(t.table.data(), 256);
            else {
                r.odd = t.function | (t.octaves << 8);
                r.color[0] = (float)t.lacunarity; r.color[1] = (float)t.persistence; r.color[2] = (float)t.power;
                r.resource = place(t.octave_tables.data(), t.octave_tables.size());
                if (t.function == RT_NOISE_TURBULENCE) r.even = place(t.table.data(), 256);
            }
            break;
        }
        case K::Image: {
            r.kind = RT_TEX_IMAGE;
            uint32_t found = UINT32_MAX;
            for (auto& e : img_cache) if (e.first == &t) found = e.second;
            if (found == UINT32_MAX) {
                found = (uint32_t)images.size();
                rt_image im{t.width, t.height, (uint64_t)texels.size()};
                images.push_back(im);
                texels.insert(texels.end(), t.rgb.begin(), t.rgb.end());
                img_cache.push_back({&t, found});
            }
            r.resource = found;
            break;
        }
    }
    uint32_t idx = (uint32_t)textures.size();
    textures.push_back(r);
    tex_cache.push_back({&t, idx});
    return idx;
}

void Scene::Impl::add_object(const Object& obj) {
    using K = Shape::Kind;
    rt_prim P{};
    P.xf_first = (uint32_t)xforms.size();
    uint32_t flips = 0;
    bool medium = false;
    const Shape::Node* cur = &obj.shape.node();
    for (;;) {
        if (cur->kind == K::Flipped) { if (!medium) ++flips; cur = cur->inner.get(); }
        else if (cur->kind == K::Translated) {
            rt_xform x{}; x.kind = RT_XF_TRANSLATE; x.a = cur->offset[0]; x.b = cur->offset[1]; x.c = cur->offset[2];
            xforms.push_back(x); cur = cur->inner.get();
        } else if (cur->kind == K::Rotated) {
            double a = cur->angle * (PI / 180.0);
            rt_xform x{}; x.kind = RT_XF_ROTATE_X + (uint32_t)cur->axis; x.a = std::cos(a); x.b = std::sin(a);
            xforms.push_back(x); cur = cur->inner.get();
        } else if (cur->kind == K::ConstantMedium) {
            if (medium) throw std::invalid_argument("Scene: a ConstantMedium nested inside another ConstantMedium is not supported");
            medium = true; P.density = (float)cur->density; P.xf_outer = (uint32_t)(xforms.size() - P.xf_first);
            if (cur->multi_scatter) P.flags |= RT_PRIM_MULTI_SCATTER;
            cur = cur->inner.get();
        } else break;
    }
    size_t n_xf = xforms.size() - P.xf_first;
    if (n_xf > RT_MAX_XFORMS_PER_PRIM) throw std::invalid_argument("Scene: more than RT_MAX_XFORMS_PER_PRIM nested Translated/Rotated wrappers");
    P.xf_count = (uint16_t)n_xf;
    if (flips & 1u) P.flags |= RT_PRIM_FLIP;
    if (medium) P.flags |= RT_PRIM_MEDIUM;
    switch (cur->kind) {
        case K::Sphere: case K::MovingSphere: {
            rt_sphere s{};
            for (int c = 0; c < 3; ++c) { s.center[c] = cur->center[c]; s.motion[c] = cur->kind == K::MovingSphere ? cur->motion[c] : 0.0; }
            s.radius = cur->radius; s.start_time = cur->start_time;
            P.geom = cur->kind == K::Sphere ? RT_GEOM_SPHERE : RT_GEOM_MOVING_SPHERE;
            P.geom_index = (uint32_t)spheres.size(); spheres.push_back(s);
            // A medium bounded by a sphere can never be hit: the second boundary query
            // (shape.rs:390) recomputes the same near root, which is not > t + 0.001.
            if (cur->both_roots) P.flags |= RT_PRIM_BOTH_ROOTS;
            else if (medium) P.flags |= RT_PRIM_INERT;
            break;
        }
        case K::Rectangle: {
            rt_rect r{}; r.axis = (uint32_t)cur->axis; r.x0 = cur->x0; r.y0 = cur->y0; r.x1 = cur->x1; r.y1 = cur->y1; r.height = cur->height;
            P.geom = RT_GEOM_RECT; P.geom_index = (uint32_t)rects.size(); rects.push_back(r);
            break;
        }
        default: {  // Box
            rt_box b{};
            for (int c = 0; c < 3; ++c) { b.min[c] = cur->bmin[c]; b.max[c] = cur->bmax[c]; }
            P.geom = RT_GEOM_BOX; P.geom_index = (uint32_t)boxes.size(); boxes.push_back(b);
            break;
        }
    }
    rt_material M{};
    using MK = materials::Material::Kind;
    const materials::Material& m = obj.material;
    M.kind = m.kind == MK::Lambertian ? RT_MAT_LAMBERTIAN : m.kind == MK::Metal ? RT_MAT_METAL : m.kind == MK::Dielectric ? RT_MAT_DIELECTRIC
             : m.kind == MK::DiffuseLight ? RT_MAT_DIFFUSE_LIGHT : RT_MAT_ISOTROPIC;
    if (m.texture) M.texture = add_texture(m.texture->node());
    for (int c = 0; c < 3; ++c) M.albedo[c] = (float)m.albedo[c];
    M.param = (float)m.param;
    if (m.face_normal) M.reserved[1] |= RT_MATF_FACE_NORMAL;
    uint32_t mi = UINT32_MAX;
    for (uint32_t k = 0; k < materials.size(); ++k) if (std::memcmp(&materials[k], &M, sizeof(M)) == 0) { mi = k; break; }
    if (mi == UINT32_MAX) { mi = (uint32_t)materials.size(); materials.push_back(M); }
    P.material = mi;
    prims.push_back(P);
}

// ---- BVH: BvhNode::create_ (lib.rs:160-200) restated -- axis = rng.gen_range(0, 3) drawn before
// recursing, stable sort by bound().min[axis], left = first n/2 -- then written out in depth-first
// pre-order with skip links.  The reference's traversal makes the left-to-right order of the leaves
// observable (see rt_abi.h), so the tree must be the reference's tree for the reference's rng stream.
namespace {
struct BuildPrim { uint32_t id; bool active, exact; double mn[3], mx[3]; };
struct Builder {
    std::vector<BuildPrim>& P;
    std::vector<rt_bvh_node>& out;
    Rng& rng;
    double pad;

    static float down(double x, double pad) { float f = (float)(x - pad); while ((double)f > x - pad) f = std::nextafterf(f, -INFINITY); return f; }
    static float up(double x, double pad) { float f = (float)(x + pad); while ((double)f < x + pad) f = std::nextafterf(f, INFINITY); return f; }

    // Emits the subtree of P[lo,hi); returns false (and emits nothing) if it holds no active primitive.
    bool build(size_t lo, size_t hi, double* mn, double* mx) {
        const size_t n = hi - lo;
        if (n == 1) {
            if (!P[lo].active) return false;
            rt_bvh_node N{};
            for (int c = 0; c < 3; ++c) {
                mn[c] = P[lo].mn[c]; mx[c] = P[lo].mx[c];
                // box-shaped primitives keep their exact bound() (see RT_BVH_EXACT in rt_abi.h); the rest are widened
                N.min[c] = P[lo].exact ? (float)mn[c] : down(mn[c], pad);
                N.max[c] = P[lo].exact ? (float)mx[c] : up(mx[c], pad);
            }
            N.prim = P[lo].id | (P[lo].exact ? RT_BVH_EXACT : 0u); N.skip = (uint32_t)out.size() + 1;
            out.push_back(N);
            return true;
        }
        const int axis = (int)(uint32_t)(rng.gen() * 3.0);
        std::stable_sort(P.begin() + lo, P.begin() + hi, [axis](const BuildPrim& a, const BuildPrim& b) { return a.mn[axis] < b.mn[axis]; });
        const size_t me = out.size();
        out.push_back(rt_bvh_node{});
        double lmn[3], lmx[3], rmn[3], rmx[3];
        const bool l = build(lo, lo + n / 2, lmn, lmx);
        const bool r = build(lo + n / 2, hi, rmn, rmx);
        if (!l && !r) { out.pop_back(); return false; }
        if (l != r) {  // one side is empty: the other side's root takes this node's place
            out.erase(out.begin() + me);
            for (size_t k = me; k < out.size(); ++k) out[k].skip -= 1;
            for (int c = 0; c < 3; ++c) { mn[c] = l ? lmn[c] : rmn[c]; mx[c] = l ? lmx[c] : rmx[c]; }
            return true;
        }
        rt_bvh_node& N = out[me];
        for (int c = 0; c < 3; ++c) {
            mn[c] = std::fmin(lmn[c], rmn[c]); mx[c] = std::fmax(lmx[c], rmx[c]);
            N.min[c] = down(mn[c], pad); N.max[c] = up(mx[c], pad);
        }
        N.prim = RT_BVH_NONE; N.skip = (uint32_t)out.size();
        return true;
    }
};
}  // namespace

Scene::Scene(std::vector<Object> objects, const geometry::Vec3& background, Rng& rng, const FlattenOptions& opt) : impl_(new Impl()) {
    if (objects.empty()) throw std::invalid_argument("Scene::new: the object list is empty");
    Impl& I = *impl_;
    for (const Object& o : objects) I.add_object(o);
    std::vector<BuildPrim> bp;
    double max_abs = 0.0;
    for (uint32_t i = 0; i < objects.size(); ++i) {
        BoundingBox b = *objects[i].shape.bound();
        BuildPrim p; p.id = i;
        p.active = opt.keep_inert || !(I.prims[i].flags & RT_PRIM_INERT);
        p.exact = I.prims[i].geom == RT_GEOM_BOX;
        for (int c = 0; c < 3; ++c) {
            p.mn[c] = b.min[c]; p.mx[c] = b.max[c];
            if (p.active && std::isfinite(b.min[c])) max_abs = std::fmax(max_abs, std::fabs(b.min[c]));
            if (p.active && std::isfinite(b.max[c])) max_abs = std::fmax(max_abs, std::fabs(b.max[c]));
        }
        bp.push_back(p);
    }
    // Boxes are widened by a few fp32 ulp of the largest coordinate in the scene so that the fp32 slab
    // test never skips a node that the reference's f64 test (shape.rs:580-599) would enter.
    const double pad = std::fmax(1e-6, 4e-6 * max_abs);
    Builder B{bp, I.bvh, rng, pad};
    double mn[3], mx[3];
    B.build(0, bp.size(), mn, mx);
    rt_scene_desc& D = I.desc;
    D.abi_version = RT_ABI_VERSION;
    D.n_prims = (uint32_t)I.prims.size(); D.prims = I.prims.data();
    D.n_spheres = (uint32_t)I.spheres.size(); D.spheres = I.spheres.data();
    D.n_rects = (uint32_t)I.rects.size(); D.rects = I.rects.data();
    D.n_boxes = (uint32_t)I.boxes.size(); D.boxes = I.boxes.data();
    D.n_xforms = (uint32_t)I.xforms.size(); D.xforms = I.xforms.data();
    D.n_materials = (uint32_t)I.materials.size(); D.materials = I.materials.data();
    D.n_textures = (uint32_t)I.textures.size(); D.textures = I.textures.data();
    D.n_images = (uint32_t)I.images.size(); D.images = I.images.data();
    D.texels = I.texels.data(); D.texel_bytes = I.texels.size();
    D.n_perm_tables = (uint32_t)(I.perm_tables.size() / 256); D.perm_tables = I.perm_tables.data();
    D.n_bvh_nodes = (uint32_t)I.bvh.size(); D.bvh = I.bvh.data();
    for (int c = 0; c < 3; ++c) D.background[c] = background[c];
}
Scene::~Scene() { if (impl_ && impl_->dev) rt_scene_destroy(impl_->dev); }
const rt_scene_desc& Scene::desc() const { return impl_->desc; }
size_t Scene::num_objects() const { return impl_->prims.size(); }
rt_scene* Scene::device_scene(int device) const {
    if (impl_->dev && impl_->dev_index == device) return impl_->dev;
    if (impl_->dev) { rt_scene_destroy(impl_->dev); impl_->dev = nullptr; }
    rt_scene* sc = nullptr;
    rt_status st = rt_scene_create(&impl_->desc, device, &sc);
    if (st != RT_OK) throw std::runtime_error(std::string("rt_scene_create failed: ") + rt_last_error());
    impl_->dev = sc; impl_->dev_index = device;
    return sc;
}

uint8_t quantise(double x) {
    const double y = std::sqrt(std::fmin(std::fmax(x, 0.0), 1.0)) * 255.999;
    if (!(y > 0.0)) return 0;
    if (y >= 255.0) return 255;
    return (uint8_t)y;
}

std::vector<uint8_t> pixels(const camera::Camera& camera, const Scene& scene, ImageOptions io, const RenderOptions& ro) {
    rt_image_options o{io.height, io.width, (uint64_t)io.samples_per_pixel, (uint64_t)io.max_reflections};
    std::vector<uint8_t> out((size_t)3 * io.width * io.height);
    if (ro.devices.size() <= 1) {
        rt_scene* sc = scene.device_scene(ro.devices.empty() ? ro.device : ro.devices[0]);
        rt_render_options r{};
        r.seed = ro.seed; r.rank = 0; r.world_size = 1;
        rt_status st = rt_render(sc, &camera.abi(), &o, &r, out.data(), nullptr);
        if (st != RT_OK) throw std::runtime_error(std::string("rt_render failed: ") + rt_last_error());
        return out;
    }
    // one thread, one device copy of the scene and one frame of means per device
    const size_t n_dev = ro.devices.size(), n = out.size();
    std::vector<std::vector<double>> means(n_dev, std::vector<double>(n));
    std::vector<std::string> errors(n_dev);
    std::vector<std::thread> pool;
    for (size_t k = 0; k < n_dev; ++k)
        pool.emplace_back([&, k]() {
            rt_scene* sc = nullptr;
            if (rt_scene_create(&scene.desc(), ro.devices[k], &sc) != RT_OK) { errors[k] = std::string("rt_scene_create failed: ") + rt_last_error(); return; }
            rt_render_options r{};
            r.seed = ro.seed; r.rank = (uint32_t)k; r.world_size = (uint32_t)n_dev;
            if (rt_render(sc, &camera.abi(), &o, &r, nullptr, means[k].data()) != RT_OK) errors[k] = std::string("rt_render failed: ") + rt_last_error();
            rt_scene_destroy(sc);
        });
    for (auto& t : pool) t.join();
    for (auto& e : errors) if (!e.empty()) throw std::runtime_error(e);
    for (size_t i = 0; i < n; ++i) {
        double sum = 0.0;
        for (size_t k = 0; k < n_dev; ++k) sum += means[k][i];
        out[i] = quantise(sum);
    }
    return out;
}

void write_ppm(const std::string& path, uint32_t width, uint32_t height, const std::vector<uint8_t>& rgb) {
    std::ofstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("write_ppm: cannot open " + path);
    f << "P6\n" << width << " " << height << "\n255\n";
    f.write((const char*)rgb.data(), (std::streamsize)rgb.size());
}

// PNG: signature, IHDR (8-bit RGB), one IDAT holding a zlib stream of stored blocks (filter byte 0 per row), IEND.
void write_png(const std::string& path, uint32_t width, uint32_t height, const std::vector<uint8_t>& rgb) {
    if (rgb.size() != (size_t)3 * width * height) throw std::runtime_error("write_png: buffer size does not match the dimensions");
    static uint32_t crc_table[256];
    static bool have_table = false;
    if (!have_table) {
        for (uint32_t n = 0; n < 256; ++n) { uint32_t c = n; for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1; crc_table[n] = c; }
        have_table = true;
    }
    auto be32 = [](std::vector<uint8_t>& v, uint32_t x) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); };
    auto chunk = [&](std::ofstream& f, const char* type, const std::vector<uint8_t>& data) {
        std::vector<uint8_t> head; be32(head, (uint32_t)data.size());
        f.write((const char*)head.data(), 4);
        uint32_t c = 0xFFFFFFFFu;
        for (int k = 0; k < 4; ++k) c = crc_table[(c ^ (uint8_t)type[k]) & 0xFFu] ^ (c >> 8);
        for (uint8_t b : data) c = crc_table[(c ^ b) & 0xFFu] ^ (c >> 8);
        f.write(type, 4);
        f.write((const char*)data.data(), (std::streamsize)data.size());
        std::vector<uint8_t> tail; be32(tail, c ^ 0xFFFFFFFFu);
        f.write((const char*)tail.data(), 4);
    };
    std::vector<uint8_t> raw;  // scanlines, each behind its filter byte
    raw.reserve(((size_t)3 * width + 1) * height);
    for (uint32_t y = 0; y < height; ++y) { raw.push_back(0); raw.insert(raw.end(), rgb.begin() + (size_t)3 * width * y, rgb.begin() + (size_t)3 * width * (y + 1)); }
    std::vector<uint8_t> z{0x78, 0x01};
    uint32_t a = 1, b = 0;  // Adler-32 of the uncompressed data
    for (size_t at = 0; at < raw.size() || at == 0;) {
        const size_t len = std::min<size_t>(65535, raw.size() - at);
        z.push_back(at + len >= raw.size() ? 1 : 0);
        z.push_back(len & 0xFF); z.push_back(len >> 8); z.push_back(~len & 0xFF); z.push_back((~len >> 8) & 0xFF);
        for (size_t k = 0; k < len; ++k) { a = (a + raw[at + k]) % 65521u; b = (b + a) % 65521u; }
        z.insert(z.end(), raw.begin() + at, raw.begin() + at + len);
        at += len;
        if (len == 0) break;
    }
    be32(z, (b << 16) | a);
    std::ofstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("write_png: cannot open " + path);
    const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    f.write((const char*)sig, 8);
    std::vector<uint8_t> ihdr; be32(ihdr, width); be32(ihdr, height);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    chunk(f, "IHDR", ihdr);
    chunk(f, "IDAT", z);
    chunk(f, "IEND", {});
}

// ----------------------------------------------------------------- examples
namespace examples {
using camera::CameraOptions;
using geometry::Axis;
using geometry::Point3;
using geometry::Vec3;
using materials::Material;
using textures::Texture;

Example random_spheres(Rng& rng) {  // examples/random_spheres.rs:23-116
    Example e;
    e.camera = CameraOptions{0.1, 16.0 / 9.0, 10.0, Point3(0, 0, 0), Point3(13, 2, 3), 0.5, 0.0, 20.0, Vec3(0, 1, 0)};
    e.objects.push_back(Object{Shape::Sphere(Point3(0, -1000, 0), 1000.0),
                               Material::Lambertian(Texture::Checkered(Texture::SolidColor(Vec3(0.9, 0.9, 0.9)), Texture::SolidColor(Vec3(0.2, 0.3, 0.1))))});
    for (int a = -11; a < 11; ++a) for (int b = -11; b < 11; ++b) {
        double cx = (double)a + 0.9 * rng.gen();
        double cz = (double)b + 0.9 * rng.gen();
        Point3 center(cx, 0.2, cz);
        if (center.dist(Point3(4.0, 0.2, 0.0)) > 0.9) {
            double roll = rng.gen();
            Point3 center2 = center;
            std::optional<Material> material;
            if (roll < 0.8) {
                center2 = center + Vec3(0.0, rng.gen_range(0.0, 0.5), 0.0);
                Vec3 c1 = Vec3::random(0.0, 1.0, rng);
                Vec3 c2 = Vec3::random(0.0, 1.0, rng);
                material = Material::Lambertian(Texture::SolidColor(c1 + c2));
            } else if (roll < 0.95) {
                Vec3 albedo = Vec3::random(0.5, 1.0, rng);
                double fuzz = rng.gen_range(0.0, 0.5);
                material = Material::Metal(albedo, fuzz);
            } else {
                material = Material::glass();
            }
            e.objects.push_back(Object{Shape::sphere(0.2, {0.0, center}, {1.0, center2}), *material});
        }
    }
    e.objects.push_back(Object{Shape::Sphere(Point3(0, 1, 0), 1.0), Material::glass()});
    e.objects.push_back(Object{Shape::Sphere(Point3(-4, 1, 0), 1.0), Material::Lambertian(Texture::SolidColor(Vec3(0.4, 0.2, 0.1)))});
    e.objects.push_back(Object{Shape::Sphere(Point3(4, 1, 0), 1.0), Material::Metal(Vec3(0.7, 0.6, 0.5), 0.0)});
    e.background = Vec3(1, 1, 1);
    return e;
}

static CameraOptions cornell_camera() {  // examples/cornell.rs:24-34
    return CameraOptions{0.0, 1.0, 10.0, Point3(278, 278, 0), Point3(278, 278, -800), 1.0, 0.0, 40.0, Vec3(0, 1, 0)};
}
static void cornell_walls(Example& e) {  // examples/cornell.rs:38-76
    auto lam = [](double r, double g, double b) { return Material::Lambertian(Texture::SolidColor(Vec3(r, g, b))); };
    e.objects.push_back(Object{Shape::rectangle(Point3(0, 0, 0), Point3(0, 555, 555)), lam(0.65, 0.05, 0.05)});
    e.objects.push_back(Object{Shape::rectangle(Point3(555, 0, 0), Point3(555, 555, 555)).flipped(), lam(0.12, 0.45, 0.15)});
    e.objects.push_back(Object{Shape::rectangle(Point3(0, 0, 0), Point3(555, 0, 555)), lam(0.73, 0.73, 0.73)});
    e.objects.push_back(Object{Shape::rectangle(Point3(0, 555, 0), Point3(555, 555, 555)).flipped(), lam(0.73, 0.73, 0.73)});
    e.objects.push_back(Object{Shape::rectangle(Point3(0, 0, 555), Point3(555, 555, 555)).flipped(), lam(0.73, 0.73, 0.73)});
}
static Shape cornell_box1() { return Shape::new_box(Point3(0, 0, 0), Point3(165, 330, 165)).rotated(Axis::Y, 15.0).translated(Vec3(265, 0, 295)); }
static Shape cornell_box2() { return Shape::new_box(Point3(0, 0, 0), Point3(165, 165, 165)).rotated(Axis::Y, -18.0).translated(Vec3(130, 0, 65)); }

Example cornell() {  // examples/cornell.rs
    Example e; e.camera = cornell_camera();
    cornell_walls(e);
    Material white = Material::Lambertian(Texture::SolidColor(Vec3(0.73, 0.73, 0.73)));
    e.objects.push_back(Object{cornell_box1(), white});
    e.objects.push_back(Object{cornell_box2(), white});
    e.objects.push_back(Object{Shape::rectangle(Point3(213, 554, 227), Point3(343, 554, 332)), Material::DiffuseLight(Texture::SolidColor(Vec3(15, 15, 15)))});
    e.background = Vec3(0, 0, 0);
    return e;
}
Example cornell_smoke() {  // examples/cornell_smoke.rs
    Example e; e.camera = cornell_camera();
    cornell_walls(e);
    e.objects.push_back(Object{Shape::ConstantMedium(cornell_box1(), 0.01), Material::Isotropic(Texture::SolidColor(Vec3(0, 0, 0)))});
    e.objects.push_back(Object{Shape::ConstantMedium(cornell_box2(), 0.01), Material::Isotropic(Texture::SolidColor(Vec3(1, 1, 1)))});
    e.objects.push_back(Object{Shape::rectangle(Point3(113, 554, 127), Point3(443, 554, 432)), Material::DiffuseLight(Texture::SolidColor(Vec3(7, 7, 7)))});
    e.background = Vec3(0, 0, 0);
    return e;
}
Example perlin() {  // examples/perlin.rs
    Example e;
    e.camera = CameraOptions{0.0, 16.0 / 9.0, 10.0, Point3(0, 0, 0), Point3(13, 2, 3), 0.01, 0.0, 20.0, Vec3(0, 1, 0)};
    noise::Perlin p;
    Texture marble = Texture::Noise(1.0, p);
    e.objects.push_back(Object{Shape::Sphere(Point3(0, -1000, 0), 1000.0), Material::Lambertian(marble)});
    e.objects.push_back(Object{Shape::Sphere(Point3(0, 2, 0), 2.0), Material::Lambertian(marble)});
    Material diff_light = Material::DiffuseLight(Texture::SolidColor(Vec3(4, 4, 4)));
    e.objects.push_back(Object{Shape::Sphere(Point3(0, 7, 0), 2.0), diff_light});
    e.objects.push_back(Object{Shape::rectangle(Point3(3, 1, -2), Point3(5, 3, -2)), diff_light});
    e.background = Vec3(0, 0, 0);
    return e;
}
// examples/final_scene.rs with its two compile errors resolved: `sphere` is cloned before its second
// use (:109/:115) and Point3::random(0.0..165.0) (:169) is three uniform draws in [0,165).
Example final_scene(Rng& rng, const Texture& earthmap) {
    Example e;
    e.camera = CameraOptions{0.1, 1.0, 10.0, Point3(0, 0, 0), Point3(13, 2, 3), 0.5, 0.0, 20.0, Vec3(0, 1, 0)};
    Material ground = Material::Lambertian(Texture::SolidColor(Vec3(0.48, 0.83, 0.53)));
    for (int i = 0; i < 20; ++i) for (int j = 0; j < 20; ++j) {
        double w = 100.0;
        double x0 = -1000.0 + (double)i * w, y0 = 0.0, z0 = -1000.0 + (double)j * w;
        double x1 = x0 + w, y1 = rng.gen_range(1.0, 101.0), z1 = z0 + w;
        e.objects.push_back(Object{Shape::new_box(Point3(x0, y0, z0), Point3(x1, y1, z1)), ground});
    }
    e.objects.push_back(Object{Shape::rectangle(Point3(123, 554, 147), Point3(423, 554, 412)), Material::DiffuseLight(Texture::SolidColor(Vec3(7, 7, 7)))});
    e.objects.push_back(Object{Shape::sphere(50.0, {0.0, Point3(400, 400, 200)}, {1.0, Point3(430, 400, 200)}), Material::Lambertian(Texture::SolidColor(Vec3(0.7, 0.3, 0.1)))});
    e.objects.push_back(Object{Shape::Sphere(Point3(260, 150, 45), 50.0), Material::glass()});
    e.objects.push_back(Object{Shape::Sphere(Point3(0, 150, 145), 50.0), Material::Metal(Vec3(0.8, 0.8, 0.9), 10.0)});
    Shape sphere = Shape::Sphere(Point3(360, 150, 145), 70.0);
    e.objects.push_back(Object{sphere, Material::glass()});
    e.objects.push_back(Object{Shape::ConstantMedium(sphere, 0.2), Material::Lambertian(Texture::SolidColor(Vec3(0.2, 0.4, 0.9)))});
    e.objects.push_back(Object{Shape::ConstantMedium(Shape::Sphere(Point3(0, 0, 0), 5000.0), 0.0001), Material::Lambertian(Texture::SolidColor(Vec3(1, 1, 1)))});
    e.objects.push_back(Object{Shape::Sphere(Point3(400, 200, 400), 100.0), Material::Lambertian(earthmap)});
    e.objects.push_back(Object{Shape::Sphere(Point3(220, 280, 300), 80.0), Material::Lambertian(Texture::Noise(0.1, noise::Perlin()))});
    Material white = Material::Lambertian(Texture::SolidColor(Vec3(0.73, 0.73, 0.73)));
    for (int k = 0; k < 1000; ++k) {
        double x = rng.gen_range(0.0, 165.0), y = rng.gen_range(0.0, 165.0), z = rng.gen_range(0.0, 165.0);
        e.objects.push_back(Object{Shape::Sphere(Point3(x, y, z), 10.0), white});
    }
    e.objects.push_back(Object{Shape::Sphere(Point3(0, -1000, 0), 1000.0),
                               Material::Lambertian(Texture::Checkered(Texture::SolidColor(Vec3(0.9, 0.9, 0.9)), Texture::SolidColor(Vec3(0.2, 0.3, 0.1))))});
    e.background = Vec3(1, 1, 1);
    return e;
}
// NOT one of the reference's examples: final_scene as the book the crate follows shows it -- seen from (478, 278, -600)
// towards (278, 278, 0) with a 40 degree field of view and a pinhole lens, every sphere also hit from within
// (RT_PRIM_BOTH_ROOTS: the glass balls refract on the way out, the two balls of smoke scatter) and every material handed the
// normal turned against the ray (RT_MATF_FACE_NORMAL).  Objects, sizes, colours and random draws are final_scene's own.  For
// comparison pictures; no parity claim hangs on it.
Example final_scene_book(Rng& rng, const Texture& earthmap) {
    Example e = final_scene(rng, earthmap);
    e.camera = CameraOptions{0.0, 1.0, 10.0, Point3(278, 278, 0), Point3(478, 278, -600), 1.0, 0.0, 40.0, Vec3(0, 1, 0)};
    for (Object& o : e.objects) {
        o.shape = o.shape.with_both_roots();
        o.material = o.material.facing_normal();
    }
    return e;
}
Example by_name(const std::string& name, Rng& rng, const Texture* earthmap) {
    if (name == "random_spheres") return random_spheres(rng);
    if (name == "cornell") return cornell();
    if (name == "perlin") return perlin();
    if (name == "cornell_smoke") return cornell_smoke();
    if (name == "final_scene" || name == "final_scene_book") {
        if (!earthmap) throw std::invalid_argument("final_scene needs the earth map texture");
        return name == "final_scene" ? final_scene(rng, *earthmap) : final_scene_book(rng, *earthmap);
    }
    throw std::invalid_argument("unknown example scene: " + name);
}
}  // namespace examples

}  // namespace raytracing
