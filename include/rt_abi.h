/* rt_abi.h -- C ABI of the MI355X path-tracing library (librt_hip.so).
 *
 * The reference crate has no FFI layer: its public Rust API is the boundary.
 * The entry points below are what a Rust facade crate (or any other host)
 * binds in place of the reference's render loop.  Each one names the
 * reference interface it replaces (paths relative to the reference checkout).
 *
 * Conventions
 *   - plain C structs, pointers and sizes; no C++ or torch types;
 *   - the library copies every array of a scene description during
 *     rt_scene_create(); the caller may free its arrays on return;
 *   - every function returns an rt_status and never aborts across the ABI;
 *     rt_last_error() returns a thread-local message for the last failure;
 *   - "device" pointers are HIP device addresses on the scene's device;
 *   - a scene handle owns device state that rendering changes (a workspace that
 *     grows with the frame, streams, the schedule learnt from earlier frames, the
 *     regrouped tree): every entry point that takes a handle holds the handle's
 *     lock for the duration of the call, so calls from several threads on one
 *     handle are safe and run one after the other.  rt_render_device returns with
 *     its work still queued; the next call on the handle queues behind it.
 *     rt_scene_destroy must not run beside any other call on the handle.
 */
#ifndef RT_ABI_H
#define RT_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 3: rt_profile grew (tail_from, two_stream_passes, pass_paths, two_bounce_passes); rt_render_options.stream orders the render with the caller's
 * stream instead of naming the launch stream; noise textures read rt_texture.odd / even / color; new RT_RENDER_* and RT_PRIM_*
 * bits.  rt_scene_create refuses a description that carries another version. */
#define RT_ABI_VERSION 3u

typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID_ARGUMENT = 1, /* null pointer, bad index, width/height < 2, empty shutter interval ... */
    RT_ERR_EMPTY_SCENE = 2,      /* Scene::new asserts !objects.is_empty() (src/lib.rs:161) */
    RT_ERR_UNSUPPORTED = 3,      /* description uses something the kernels do not implement */
    RT_ERR_NO_DEVICE = 4,        /* no usable HIP device */
    RT_ERR_HIP = 5,              /* a HIP runtime call failed; see rt_last_error() */
    RT_ERR_OUT_OF_MEMORY = 6
} rt_status;

/* ---------------------------------------------------------------- scene
 * Flattened form of `Scene { objects: BvhNode<Object>, background_color }`
 * (src/lib.rs:119-133): one rt_prim per `Object` (src/lib.rs:261-266), in the
 * order of the `Vec<Object>` given to Scene::new; the index in that list is the
 * primitive id used for RNG keying and reported in hit records.
 */

/* Innermost geometry of a `Shape` (src/geometry/shape.rs:62-106) once the
 * Flipped / Translated / Rotated / ConstantMedium wrappers are peeled off. */
enum { RT_GEOM_SPHERE = 0, RT_GEOM_MOVING_SPHERE = 1, RT_GEOM_RECT = 2, RT_GEOM_BOX = 3 };

enum {
    RT_PRIM_FLIP = 1u,   /* odd number of Shape::Flipped wrappers (outside the medium, if any) */
    RT_PRIM_MEDIUM = 2u, /* Shape::ConstantMedium { boundary, density } */
    RT_PRIM_INERT = 4u,  /* provably never hit (a medium whose boundary is a sphere: the second
                            boundary query of shape.rs:390 repeats the near root); kept in the
                            list so primitive ids stay stable, left out of the BVH */
    RT_PRIM_BOTH_ROOTS = 8u, /* NOT the reference's behaviour (a variant for comparison with the book the crate follows): a
                            sphere whose near root lies outside the ray interval is tried at its far root as well, so a ray that
                            starts inside a sphere hits it from within (glass refracts on the way out, a medium bounded by a
                            sphere scatters).  shape.rs:262-271 only ever takes the near root */
    RT_PRIM_MULTI_SCATTER = 16u /* NOT the reference's behaviour either (media only): the two boundary queries of the medium
                            (shape.rs:388-390) evaluate the ray behind its origin where it is instead of at the origin (Ray::at
                            clamps t <= 0 to the origin, geometry.rs:340-346), so a ray that starts inside the boundary finds
                            the way in behind it and the way out ahead, and a path scatters inside the medium as often as its
                            free flights end there.  The reference's media scatter at most once per entry: from inside, the
                            sides behind the origin all pass the extents test at the origin and both queries come back negative */
};

typedef struct rt_prim {
    uint8_t geom;       /* RT_GEOM_* ; index space of `geom_index` */
    uint8_t flags;      /* RT_PRIM_* */
    uint16_t xf_count;  /* rigid wrappers, outermost first */
    uint32_t xf_first;  /* first element in rt_scene_desc.xforms */
    uint32_t geom_index;
    uint32_t material;
    float density;      /* medium only */
    uint32_t xf_outer;  /* medium only: how many of the wrappers sit outside the ConstantMedium */
    uint32_t reserved[2];
} rt_prim; /* 32 bytes */

/* Sphere { center, radius } and MovingSphere { ray, radius, start_time, .. }
 * (shape.rs:64-73).  Kept in f64: the discriminant half_b^2 - a*c (shape.rs:265)
 * cancels catastrophically in fp32 for the radius-1000 spheres of the examples. */
typedef struct rt_sphere {
    double center[3]; /* MovingSphere: ray.origin = centre at start_time */
    double radius;
    double motion[3]; /* MovingSphere: ray.direction = end_center - start_center; else 0 */
    double start_time;
} rt_sphere; /* 64 bytes */

/* Rectangle { axis, lower_left, upper_right, height } (shape.rs:77-82).  Geometry crosses the boundary in the
 * reference's f64; the library keeps each coordinate as an fp32 value plus an fp32 remainder, so that the
 * difference (plane - ray origin), which sets t for a ray starting close to the plane, is formed without
 * the rounding of the plane to fp32 (1e-5 of t for grazing rays otherwise). */
typedef struct rt_rect {
    uint32_t axis; /* 0 X, 1 Y, 2 Z: the axis orthogonal to the rectangle */
    uint32_t reserved;
    double x0, y0, x1, y1, height;
} rt_rect; /* 48 bytes */

/* Box { min, max, sides } (shape.rs:88-92); the six sides are those of
 * Shape::new_box (shape.rs:157-215), in that order. */
typedef struct rt_box {
    double min[3];
    double max[3];
} rt_box; /* 48 bytes */

/* One rigid wrapper: Translated { offset } (shape.rs:98) or Rotated { axis,
 * angle } (shape.rs:101-105) with cos/sin of +angle precomputed.  Kept in f64: a ray is
 * moved into the shape's frame in f64 (see rt_core.h). */
enum { RT_XF_TRANSLATE = 0, RT_XF_ROTATE_X = 1, RT_XF_ROTATE_Y = 2, RT_XF_ROTATE_Z = 3 };
typedef struct rt_xform {
    uint32_t kind;
    uint32_t reserved;
    double a, b, c; /* translate: offset xyz; rotate: a = cos(angle), b = sin(angle), c unused */
} rt_xform;         /* 32 bytes */

/* Material (src/materials.rs:8-18). */
enum { RT_MAT_LAMBERTIAN = 0, RT_MAT_METAL = 1, RT_MAT_DIELECTRIC = 2, RT_MAT_DIFFUSE_LIGHT = 3, RT_MAT_ISOTROPIC = 4 };
typedef struct rt_material {
    uint32_t kind;
    uint32_t texture; /* Lambertian / DiffuseLight / Isotropic */
    float albedo[3];  /* Metal / Dielectric */
    float param;      /* Metal: fuzz; Dielectric: refraction_index */
    uint32_t reserved[2]; /* [0]: 0 (the library keeps notes there on its own copy); [1]: RT_MATF_* */
} rt_material; /* 32 bytes */
/* NOT the reference's behaviour (a variant like RT_PRIM_BOTH_ROOTS): the material is handed the normal turned against the
 * incoming ray, as the book's set_face_normal does; which face was hit (IntersectionPoint::face, shape.rs:552-558) is still
 * decided on the shape's own normal.  The reference never flips (materials.rs:28-113 use p.normal as it comes). */
#define RT_MATF_FACE_NORMAL 1u

/* Texture (src/textures.rs:8-19). */
enum { RT_TEX_SOLID = 0, RT_TEX_CHECKERED = 1, RT_TEX_NOISE = 2, RT_TEX_IMAGE = 3 };
/* Noise { scale, noise } (textures.rs:14-17) holds an open `dyn NoiseFn`; the examples only ever put noise::Perlin there
 * (perlin.rs:40, final_scene.rs:160).  The device evaluates three functions of the noise crate's family, chosen by the low
 * byte of rt_texture.odd of a noise texture (0, the value every description written before these existed carries, is
 * Perlin).  Fbm and Turbulence are extensions beyond what the reference's examples use: no reference image depends on them.
 *   RT_NOISE_PERLIN      gradient noise over permutation table `resource`;
 *   RT_NOISE_FBM         sum over octaves i < n of  persistence^i * Perlin_i(point * lacunarity^i),  Perlin_i over table
 *                        resource + i (the crate's Fbm seeds its octave i with seed + i);
 *   RT_NOISE_TURBULENCE  Perlin over table `even` at the point displaced, per coordinate c in (x, y, z), by
 *                        power * Fbm_c(point + offset_c): Fbm_c has `n` octaves over tables resource + c + i (the crate's
 *                        Turbulence seeds its three distortion functions with seed, seed + 1, seed + 2), offset_c are the
 *                        crate's fixed fractions k / 65536.
 * n = bits 8..15 of `odd` (1..32); lacunarity, persistence, power = color[0], color[1], color[2]. */
enum { RT_NOISE_PERLIN = 0, RT_NOISE_FBM = 1, RT_NOISE_TURBULENCE = 2 };
typedef struct rt_texture {
    uint32_t kind;
    float color[3];    /* SolidColor; Noise: lacunarity, persistence, power (Fbm / Turbulence) */
    uint32_t odd, even;/* Checkered: indices of the two child textures; Noise: function | octaves << 8, source table (Turbulence) */
    float scale;       /* Noise */
    uint32_t resource; /* Noise: (first) permutation table index; Image: image index */
} rt_texture;          /* 32 bytes */

/* Image(RgbImage) (textures.rs:18): 8-bit RGB, row 0 at the top, 3 bytes per texel. */
typedef struct rt_image {
    uint32_t width, height;
    uint64_t texel_offset; /* byte offset into rt_scene_desc.texels */
} rt_image;

/* Linearised BVH in depth-first pre-order with skip links (stackless traversal).
 *
 * The reference's BvhNode::scatter_ (src/lib.rs:228-257) visits left then right with a
 * shrinking tmax and tests each node's box first, so what it computes is a scan over the
 * LEAVES IN DEPTH-FIRST ORDER: a leaf's object is tried iff the ray overlaps the leaf's box
 * within (tmin, current tmax); an accepted hit lowers tmax.  That order is observable: the
 * y-sides of Shape::new_box carry their x and z extents swapped (shape.rs:180-195 against
 * :315-328), so a box can be hit outside its own bounding box, and whether such a hit counts
 * depends on the leaf test and on the hits found before it.  The host therefore builds the
 * tree exactly as BvhNode::create_ does (src/lib.rs:160-200: random axis, stable sort by
 * bound().min[axis], split at n/2) and the kernels walk it in the same order.
 *
 * Node i's subtree is nodes [i+1, skip); a leaf has prim != RT_BVH_NONE and skip == i+1.
 * Inner boxes must contain the boxes of the leaves below them (they may be wider).  Inert
 * primitives are left out.
 *
 * A leaf whose prim field carries RT_BVH_EXACT holds the primitive's exact bound() box and is
 * tested the way BoundingBox::hit does it (shape.rs:580-599: t = (bound - origin) / direction,
 * miss when tmax <= tmin).  The host sets it on every box-shaped primitive: their leaf test is
 * observable (hits outside the box; the strict test is what decides between two boxes that
 * share a face).  All other boxes are widened and tested conservatively. */
#define RT_BVH_NONE 0xFFFFFFFFu
#define RT_BVH_EXACT 0x80000000u
typedef struct rt_bvh_node {
    float min[3];
    uint32_t skip; /* index of the next node in pre-order that is not in this node's subtree */
    float max[3];
    uint32_t prim; /* leaf: primitive id (| RT_BVH_EXACT); inner node: RT_BVH_NONE */
} rt_bvh_node;     /* 32 bytes */

typedef struct rt_scene_desc {
    uint32_t abi_version; /* RT_ABI_VERSION */
    /* n_prims: at least 1 (Scene::new asserts a non-empty object list, src/lib.rs:161) and below 2^21 */
    uint32_t n_prims, n_spheres, n_rects, n_boxes, n_xforms, n_materials, n_textures, n_images, n_perm_tables, n_bvh_nodes;
    uint32_t reserved0;
    const rt_prim* prims;
    const rt_sphere* spheres;
    const rt_rect* rects;
    const rt_box* boxes;
    const rt_xform* xforms;
    const rt_material* materials;
    const rt_texture* textures;
    const rt_image* images;
    const uint8_t* texels;
    uint64_t texel_bytes;
    const uint8_t* perm_tables; /* n_perm_tables x 256 bytes */
    const rt_bvh_node* bvh;
    double background[3]; /* Scene.background_color */
} rt_scene_desc;

#define RT_MAX_XFORMS_PER_PRIM 8u
#define RT_MAX_TEXTURE_NESTING 8u

/* `Camera` (src/lib.rs:314-341) after Camera::new (src/lib.rs:347-384), computed by the host in f64. */
typedef struct rt_camera {
    double origin[3], lower_left_corner[3], horizontal[3], vertical[3];
    double u[3], v[3]; /* onb[0], onb[1] */
    double lens_radius, shutter_open, shutter_close;
} rt_camera;

/* `ImageOptions` (src/lib.rs:108-117), same four fields. */
typedef struct rt_image_options {
    uint32_t height, width;
    uint64_t samples_per_pixel, max_reflections;
} rt_image_options;

/* No counterpart in the reference (it has neither seeds nor devices). */
typedef struct rt_render_options {
    uint64_t seed;            /* keys the Philox streams (pixel, sample, bounce, stream) */
    uint32_t rank, world_size;/* this process renders the tiles with tile_id % world_size == rank */
    uint32_t tile_w, tile_h;  /* 0 = default 16 x 16 */
    uint32_t samples_per_pass;/* 0 = automatic; does not change the result */
    uint32_t flags;           /* RT_RENDER_* */
    void* stream;             /* hipStream_t the render is ordered with: it starts after the work queued there so far, and
                                 work queued there afterwards starts after the frame is complete (the kernels themselves run
                                 on the library's own streams).  NULL = no ordering with any caller stream: the render runs on
                                 the library's own (non-blocking) streams alone, the null stream included in "no ordering";
                                 a caller that works on the legacy default stream, whose handle is also 0, passes a stream of
                                 its own and orders the two with events.  The stream must have default priority (RT_ERR_UNSUPPORTED
                                 otherwise): a stream of raised priority that waits for the library's streams is never released
                                 on this runtime */
} rt_render_options;
enum {
    RT_RENDER_PROFILE = 1u,  /* time every kernel launch with HIP events (rt_get_profile) */
    RT_RENDER_NO_TAIL = 2u,  /* every bounce as separate traverse / shade launches (same image; for tests) */
    RT_RENDER_TAIL_ONLY = 4u, /* every path runs start to end in the tail kernel (same image; for tests) */
    RT_RENDER_NODES_IN_GLOBAL = 8u, /* walk the BVH in global memory even when it would fit in LDS (same image; for tests) */
    RT_RENDER_ONE_STREAM = 16u, /* never split a pass into two halves on two streams (same image; for tests and clean kernel timings) */
    RT_RENDER_ONE_KERNEL = 64u,  /* walk and shade a bounce in one kernel whatever the tree's size (same image; for tests) */
    RT_RENDER_TWO_KERNELS = 128u,/* never do that: every bounce as a traversal launch and a shading launch (same image; for tests) */
    RT_RENDER_RECALIBRATE = 32u, /* measure the frame's schedule again (where the tail kernel takes over, whether two streams
                                   pay) instead of reusing what an earlier frame of the same shape on this handle learnt */
    RT_RENDER_KEEP_TREE = 256u  /* do not regroup the inner nodes of the library's copy of the tree by the rays of this frame
                                   (see rt_profile.tree_tests_sampled; same image either way; for tests and comparisons) */
};

typedef struct rt_scene rt_scene; /* opaque */

/* Replaces Scene::new(objects, background, rng) (src/lib.rs:127-132): uploads the
 * flattened scene to HIP device `device` (-1 = current device). */
rt_status rt_scene_create(const rt_scene_desc* desc, int device, rt_scene** out);
void rt_scene_destroy(rt_scene* scene);

/* Replaces `pixels(&camera, &scene, image_options) -> Vec<u8>` (src/lib.rs:18-32).
 * rgb_out: 3*width*height bytes, caller-owned host memory, row-major, top row first.
 * mean_out (optional): 3*width*height doubles, mean linear radiance per channel before
 * the clamp / sqrt / *255.999 quantisation of src/lib.rs:68-72.
 * With world_size > 1 only the pixels of this rank's tiles are rendered; the others are
 * written as 0 (sum the outputs of all ranks, then quantise with rt_quantise). */
rt_status rt_render(rt_scene* scene, const rt_camera* camera, const rt_image_options* image_options,
                    const rt_render_options* render_options, uint8_t* rgb_out, double* mean_out);

/* Same computation with the result left on the device: d_mean_out is a device buffer of
 * 3*width*height doubles (pixels of other ranks' tiles are written as 0).  Returns after
 * the work is enqueued on the stream; the caller synchronises. */
rt_status rt_render_device(rt_scene* scene, const rt_camera* camera, const rt_image_options* image_options,
                           const rt_render_options* render_options, double* d_mean_out);

/* rt_render_device for callers that cannot allocate (code under HIP graph capture, serving loops, harnesses that forbid
 * hipMalloc).  No counterpart in the reference.  Same arguments, same bytes in d_mean_out; calls no hipMalloc / hipFree /
 * hipHostMalloc and creates no events: its queues, hit records, sample values, per-pixel sums and counters live in one
 * static device array of the library's image (about 38.5 MB on every GPU the library is loaded on, held until the
 * process ends).  Limits, checked up front (RT_ERR_UNSUPPORTED beyond them): at most 262 144 pixels owned by this rank
 * (512 x 512 on one rank) and max_reflections <= 254; passes hold at most 262 144 paths.  RT_RENDER_PROFILE is refused.
 * All calls on one GPU share that GPU's array, whichever scene handle they use: calls whose device work overlaps on a GPU
 * race (every call overwrites what it reads before reading it, so nothing needs clearing between calls).  Calls on one
 * handle queue one behind the other on the handle's stream.  Calls on different handles run on different streams: the
 * caller orders them by giving both the same `stream` (the second call then starts after the first has handed that
 * stream back) or by synchronising in between; with stream = NULL nothing orders them.
 * Under capture of `stream` the call adds the whole render to the graph (the library's own stream joins the capture through
 * the events the call records on either side) and runs nothing; replays of the graph share the array like calls do.
 * Returns after the work is enqueued; the caller synchronises. */
rt_status rt_render_device_noalloc(rt_scene* scene, const rt_camera* camera, const rt_image_options* image_options,
                                   const rt_render_options* render_options, double* d_mean_out);

/* The per-pixel quantisation of src/lib.rs:68-72 ("max(0).min(1).sqrt() * 255.999 as u8")
 * on a device buffer of n doubles -> n bytes (device), enqueued on `stream`. */
rt_status rt_quantise_device(rt_scene* scene, const double* d_mean, uint64_t n, uint8_t* d_rgb_out, void* stream);

/* Parity hook for `Intersection::intersect(&self, ray, (tmin, tmax), time)`
 * (src/geometry/shape.rs:14) and for the closest hit of BvhNode::scatter_
 * (src/lib.rs:228-257). */
typedef struct rt_ray {
    float origin[3];
    float direction[3];
    float time;
} rt_ray;
typedef struct rt_hit { /* IntersectionPoint, shape.rs:539-550 */
    int32_t hit;
    uint32_t prim;
    float t;
    float point[3], normal[3], in_vec[3];
    float u, v;
} rt_hit;
/* scope < 0: closest hit over the whole scene through the BVH; scope >= 0: that primitive
 * alone.  medium_u (optional, n floats): the uniform variate every ConstantMedium uses
 * for ray i; when NULL it is Philox(seed; pixel = i, sample 0, bounce 0, stream 1 + prim).
 * rays / medium_u / out are host pointers. */
rt_status rt_intersect_batch(rt_scene* scene, int64_t scope, const rt_ray* rays, uint64_t n, float tmin, float tmax,
                             uint64_t seed, const float* medium_u, rt_hit* out);

/* Parity hook for `Material::scatter(&self, p, rng) -> RayHit` (src/materials.rs:28-113) together with the
 * texture look-up `Texture::color_at` (src/textures.rs:46-78): one shading event per hit record, run by the
 * same device code the renderer shades with. */
typedef struct rt_scatter { /* RayHit, src/lib.rs:300-306 */
    int32_t scattered;      /* 1 = RayHit::Scattered { ray, attenuation }, 0 = RayHit::Emitted(colour) */
    float color[3];         /* the attenuation, or the emitted colour */
    float origin[3], direction[3]; /* the scattered ray; unspecified when nothing is scattered */
} rt_scatter;
/* hits[i].prim names the object whose material scatters (hits[i].hit is ignored).  keys (optional, 3 per
 * record): the pixel, sample and bounce whose Philox streams supply the draws; NULL = (i, 0, 0).
 * hits / keys / out are host pointers. */
rt_status rt_scatter_batch(rt_scene* scene, const rt_hit* hits, uint64_t n, uint64_t seed, const uint32_t* keys, rt_scatter* out);

/* Camera rays exactly as pixel() draws them (src/lib.rs:55-60, :389-401) for n
 * (column i, row-from-bottom j, sample) triples; host pointers. */
rt_status rt_camera_rays(rt_scene* scene, const rt_camera* camera, const rt_image_options* image_options, uint64_t seed,
                         uint64_t n, const uint32_t* px_i, const uint32_t* px_j, const uint32_t* sample, rt_ray* out);

/* Kernel timings of the last rt_render / rt_render_device call made with RT_RENDER_PROFILE. */
enum { RT_K_RAYGEN = 0, RT_K_TRAVERSE = 1, RT_K_SHADE = 2, RT_K_TAIL = 3, RT_K_RESOLVE = 4, RT_K_FINALIZE = 5, RT_K_COUNT = 6 };
typedef struct rt_profile {
    double kernel_ms[RT_K_COUNT];      /* sum of launch durations, HIP events on the launch stream */
    uint64_t kernel_launches[RT_K_COUNT];
    uint64_t samples;                  /* camera paths started */
    uint64_t segments;                 /* path segments traced (ray queries) */
    uint64_t traverse_segments;        /* segments handled by the stand-alone traverse kernel */
    uint64_t passes;
    /* The first sizeable pass rendered on a scene handle lends a few hundred of its rays to the library, which regroups the
     * inner nodes of its copy of the tree so that those rays meet fewer boxes (leaves, their order and so every hit stay as
     * they are: inner nodes only cull).  These two say what that did, per sampled ray, since the handle was made: box tests
     * with the tree grouped by box areas and with the tree in use now (equal, or both 0, while nothing has been regrouped). */
    double tree_tests_by_area, tree_tests_sampled;
    /* The schedule of the last pass of the last render call (set by every call, profiled or not): the bounce from which the
     * tail kernel ran every remaining segment (max_reflections if it never did), how many passes of the call ran as two
     * halves on two streams, and the number of paths of a full pass.  The schedule is learnt per (frame shape, pass size).
     * two_bounce_passes: passes of the call whose first launch ran bounces 0 and 1 in the same lanes (scenes of a few leaves). */
    int32_t tail_from;
    uint32_t two_stream_passes;
    uint64_t pass_paths;
    uint32_t two_bounce_passes;
    uint32_t reserved;
} rt_profile;
rt_status rt_get_profile(rt_scene* scene, rt_profile* out);

/* Number of HIP devices visible to the library (0 when there is none). */
int rt_device_count(void);
const char* rt_last_error(void);
uint32_t rt_abi_version(void);
/* Paths (owned pixels x samples) the library puts in one pass when rt_render_options.samples_per_pass is 0; a frame of fewer
 * paths is one pass.  No counterpart in the reference; for callers that size warm-up frames (the schedule is learnt per pass size). */
uint64_t rt_auto_pass_paths(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_ABI_H */
