// c_api.cpp -- flat C entry points over the C++ host API (include/raytracing.hpp) for
// bindings that cannot use C++ types (the Python test harness, a future Rust facade).
// Handles are heap objects owned by the caller; every *_free releases one handle.
// Functions that can fail return NULL / non-zero and leave a message in rth_last_error().

#include <cstring>
#include <string>

#include "raytracing.hpp"

using namespace raytracing;
using geometry::Axis;
using geometry::Point3;
using geometry::Shape;
using geometry::Vec3;
using materials::Material;
using textures::Texture;

static thread_local std::string g_err;
static Point3 p3(const double* p) { return Point3(p[0], p[1], p[2]); }
static Vec3 v3(const double* p) { return Vec3(p[0], p[1], p[2]); }

struct rth_camera_options { double aperture, aspect_ratio, focus_distance, looking_at[3], origin[3], shutter_close, shutter_open, vfov, vup[3]; };

static camera::CameraOptions cam_opts(const rth_camera_options* o) {
    return camera::CameraOptions{o->aperture, o->aspect_ratio, o->focus_distance, p3(o->looking_at), p3(o->origin), o->shutter_close, o->shutter_open, o->vfov, v3(o->vup)};
}

extern "C" {

const char* rth_last_error(void) { return g_err.c_str(); }

void* rth_sphere(const double* c, double r) { return new Shape(Shape::Sphere(p3(c), r)); }
void* rth_sphere_both_roots(const double* c, double r) { return new Shape(Shape::SphereBothRoots(p3(c), r)); }
void* rth_sphere_timed(double r, double t0, const double* c0, double t1, const double* c1) {
    try { return new Shape(Shape::sphere(r, {t0, p3(c0)}, {t1, p3(c1)})); } catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}
void* rth_rectangle(const double* ll, const double* ur) { return new Shape(Shape::rectangle(p3(ll), p3(ur))); }
void* rth_new_box(const double* mn, const double* mx) { return new Shape(Shape::new_box(p3(mn), p3(mx))); }
void* rth_flipped(void* s) { return new Shape(((Shape*)s)->flipped()); }
void* rth_translated(void* s, const double* off) { return new Shape(((Shape*)s)->translated(v3(off))); }
void* rth_rotated(void* s, int axis, double angle) { return new Shape(((Shape*)s)->rotated((Axis)axis, angle)); }
void* rth_constant_medium(void* s, double density) { return new Shape(Shape::ConstantMedium(*(Shape*)s, density)); }
void* rth_constant_medium_multi(void* s, double density) { return new Shape(Shape::ConstantMediumMultiScatter(*(Shape*)s, density)); }
void rth_shape_free(void* s) { delete (Shape*)s; }
void rth_shape_bound(void* s, double* mn, double* mx) {
    geometry::BoundingBox b = *((Shape*)s)->bound();
    for (int c = 0; c < 3; ++c) { mn[c] = b.min[c]; mx[c] = b.max[c]; }
}

void* rth_tex_solid(const double* c) { return new Texture(Texture::SolidColor(v3(c))); }
void* rth_tex_checkered(void* odd, void* even) { return new Texture(Texture::Checkered(*(Texture*)odd, *(Texture*)even)); }
void* rth_tex_noise(double scale, const uint8_t* perm256) {
    std::array<uint8_t, 256> t; std::memcpy(t.data(), perm256, 256);
    return new Texture(Texture::Noise(scale, noise::Perlin(t)));
}
void* rth_tex_noise_fbm(double scale, uint32_t seed, uint32_t octaves, double lacunarity, double persistence) {
    noise::Fbm f; f.seed = seed; f.octaves = octaves; f.lacunarity = lacunarity; f.persistence = persistence;
    try { return new Texture(Texture::Noise(scale, f)); } catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}
void* rth_tex_noise_turbulence(double scale, const uint8_t* source_perm256, uint32_t seed, uint32_t roughness, double power, double lacunarity, double persistence) {
    std::array<uint8_t, 256> t; std::memcpy(t.data(), source_perm256, 256);
    noise::Turbulence n{noise::Perlin(t), seed, roughness, power, lacunarity, persistence};
    try { return new Texture(Texture::Noise(scale, n)); } catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}
void* rth_tex_image(uint32_t w, uint32_t h, const uint8_t* rgb) {
    try { return new Texture(Texture::Image(w, h, std::vector<uint8_t>(rgb, rgb + (size_t)3 * w * h))); } catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}
void* rth_tex_image_file(const char* path) {
    try { return new Texture(Texture::image(path)); } catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}
void rth_tex_free(void* t) { delete (Texture*)t; }

void* rth_mat_face_normal(void* m) { return new Material(((Material*)m)->facing_normal()); }
void* rth_mat_lambertian(void* t) { return new Material(Material::Lambertian(*(Texture*)t)); }
void* rth_mat_metal(const double* a, double fuzz) { return new Material(Material::Metal(v3(a), fuzz)); }
void* rth_mat_dielectric(const double* a, double ri) { return new Material(Material::Dielectric(v3(a), ri)); }
void* rth_mat_diffuse_light(void* t) { return new Material(Material::DiffuseLight(*(Texture*)t)); }
void* rth_mat_isotropic(void* t) { return new Material(Material::Isotropic(*(Texture*)t)); }
void rth_mat_free(void* m) { delete (Material*)m; }

void rth_perm_table(uint32_t seed, uint8_t* out256) { noise::Perlin p(seed); std::memcpy(out256, p.table.data(), 256); }

// Scene::new(objects, background, rng); keep_inert: see FlattenOptions.
void* rth_scene_new(size_t n, void** shapes, void** mats, const double* background, uint64_t seed, int keep_inert) {
    try {
        std::vector<Object> objs;
        for (size_t i = 0; i < n; ++i) objs.push_back(Object{*(Shape*)shapes[i], *(Material*)mats[i]});
        Rng rng(seed);
        FlattenOptions fo; fo.keep_inert = keep_inert != 0;
        return new Scene(std::move(objs), v3(background), rng, fo);
    } catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}
void rth_scene_free(void* sc) { delete (Scene*)sc; }
const rt_scene_desc* rth_scene_desc(void* sc) { return &((Scene*)sc)->desc(); }

// One of the five example scenes; img_* is the decoded earth map (final_scene only).
void* rth_example_scene(const char* name, uint64_t scene_seed, uint32_t img_w, uint32_t img_h, const uint8_t* img_rgb, int keep_inert,
                        rth_camera_options* cam_out) {
    try {
        Rng rng(scene_seed);
        std::optional<Texture> earth;
        if (img_rgb) earth = Texture::Image(img_w, img_h, std::vector<uint8_t>(img_rgb, img_rgb + (size_t)3 * img_w * img_h));
        examples::Example e = examples::by_name(name, rng, earth ? &*earth : nullptr);
        if (cam_out) {
            const camera::CameraOptions& c = e.camera;
            cam_out->aperture = c.aperture; cam_out->aspect_ratio = c.aspect_ratio; cam_out->focus_distance = c.focus_distance;
            for (int k = 0; k < 3; ++k) { cam_out->looking_at[k] = c.looking_at[k]; cam_out->origin[k] = c.origin[k]; cam_out->vup[k] = c.vup[k]; }
            cam_out->shutter_close = c.shutter_close; cam_out->shutter_open = c.shutter_open; cam_out->vfov = c.vfov;
        }
        FlattenOptions fo; fo.keep_inert = keep_inert != 0;
        return new Scene(std::move(e.objects), e.background, rng, fo);
    } catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}

// Camera::new(options) -> rt_camera
int rth_camera_new(const rth_camera_options* o, rt_camera* out) {
    try { *out = camera::Camera(cam_opts(o)).abi(); return 0; } catch (const std::exception& e) { g_err = e.what(); return 1; }
}

// pixels(&camera, &scene, image_options) through the C++ entry (device -1 = current)
int rth_pixels(void* sc, const rth_camera_options* o, uint32_t width, uint32_t height, uint64_t spp, uint64_t max_reflections, uint64_t seed, int device,
               uint8_t* rgb_out) {
    try {
        camera::Camera cam(cam_opts(o));
        RenderOptions ro; ro.seed = seed; ro.device = device;
        std::vector<uint8_t> px = pixels(cam, *(Scene*)sc, ImageOptions{height, width, (size_t)spp, (size_t)max_reflections}, ro);
        std::memcpy(rgb_out, px.data(), px.size());
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return 1; }
}

// pixels() with the frame split over several device contexts (RenderOptions::devices)
int rth_pixels_devices(void* sc, const rth_camera_options* o, uint32_t width, uint32_t height, uint64_t spp, uint64_t max_reflections, uint64_t seed,
                       uint32_t n_devices, const int* devices, uint8_t* rgb_out) {
    try {
        camera::Camera cam(cam_opts(o));
        RenderOptions ro; ro.seed = seed; ro.devices.assign(devices, devices + n_devices);
        std::vector<uint8_t> px = pixels(cam, *(Scene*)sc, ImageOptions{height, width, (size_t)spp, (size_t)max_reflections}, ro);
        std::memcpy(rgb_out, px.data(), px.size());
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return 1; }
}

// write_ppm / write_png of the host API (path ending in .png selects PNG)
int rth_write_image(const char* path, uint32_t width, uint32_t height, const uint8_t* rgb) {
    try {
        const std::string p(path);
        std::vector<uint8_t> px(rgb, rgb + (size_t)3 * width * height);
        if (p.size() >= 4 && p.compare(p.size() - 4, 4, ".png") == 0) write_png(p, width, height, px); else write_ppm(p, width, height, px);
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return 1; }
}

}  // extern "C"
