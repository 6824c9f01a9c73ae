// rt_twin.cpp -- TEST INFRASTRUCTURE ONLY.  A g++ build of the per-ray functions of
// raytracing_amd/csrc/rt_core.h (the same source the HIP kernels inline) driven by plain
// host loops, so that intersection / shading / camera logic can be unit-tested against
// the oracle on a machine without a GPU.  Nothing in the product loads this library.

#include <algorithm>
#include <atomic>
#include <cstring>
#include <thread>
#include <vector>

#include <cstdint>
struct TwinCounters { unsigned long long nodes, leaves, prim_tests; uint32_t* trace; uint32_t trace_n, trace_cap; };
static thread_local TwinCounters g_tc;
static inline void twin_count_nodes() { g_tc.nodes++; }
static inline void twin_count_leaves() { g_tc.leaves++; if (g_tc.trace && g_tc.trace_n < g_tc.trace_cap) g_tc.trace[g_tc.trace_n++] = (uint32_t)g_tc.nodes; }
static inline void twin_count_prim_tests() { g_tc.prim_tests++; }
static unsigned long long g_generic_leaves = 0;  // leaves that took the general primitive test (single-threaded statistics runs only)
static inline void twin_count_generic_leaves() { g_generic_leaves++; }
#define RT_COUNT(name) twin_count_##name()
#include "../../raytracing_amd/csrc/rt_core.h"

using namespace rt;

namespace {
// Sample rays for the regrouping of the inner nodes (see SampleRays in rt_core.h); none = areas.
SampleRays g_sample_rays;
// The scene as the kernels see it: the caller's arrays plus the walk's own node array.
struct TwinScene {
    SceneView V{};
    std::vector<WalkNode> nodes;
    std::vector<DevRect> rects;
    std::vector<DevBox> boxes;
    std::vector<ShadeRec> shade;
    // with_nodes = false: for callers that never walk the tree (the node array takes some milliseconds to make)
    explicit TwinScene(const rt_scene_desc* D, bool with_nodes = true) {
        make_device_shapes(D, rects, boxes);
        V.prims = D->prims; V.spheres = D->spheres; V.rects = rects.data(); V.boxes = boxes.data(); V.xforms = D->xforms;
        make_shade_records(D, shade);
        V.side_rounding = scene_needs_side_rounding(D) ? 1u : 0u;
        V.fp32_planes = scene_tree_holds_a_medium(D) ? 0u : 1u;
        V.multi_scatter = scene_has_multi_scatter_media(D) ? 1u : 0u;
        V.shade = shade.data(); V.textures = D->textures; V.images = D->images; V.texels = D->texels; V.perm = D->perm_tables;
        if (with_nodes) make_walk_nodes(D, nodes, g_sample_rays.size() ? &g_sample_rays : nullptr);
        V.bvh = nodes.data();
        V.n_prims = D->n_prims; V.n_bvh_nodes = with_nodes ? (uint32_t)nodes.size() - 1 : 0;
        for (int c = 0; c < 3; ++c) V.background[c] = (float)D->background[c];
    }
    TwinScene(const TwinScene&) = delete;
};
}  // namespace

extern "C" {

void twin_intersect_batch(const rt_scene_desc* D, int64_t scope, const rt_ray* rays, uint64_t n, float tmin, float tmax, uint64_t seed,
                          const float* medium_u, rt_hit* out) {
    const TwinScene scene(D, scope < 0); const SceneView& S = scene.V;
    for (uint64_t i = 0; i < n; ++i) {
        Ray r; r.o = v3(rays[i].origin[0], rays[i].origin[1], rays[i].origin[2]);
        r.d = v3(rays[i].direction[0], rays[i].direction[1], rays[i].direction[2]); r.time = rays[i].time;
        DrawKey key; key.k0 = (uint32_t)seed; key.k1 = (uint32_t)(seed >> 32); key.pixel = (uint32_t)i; key.sample = 0; key.bounce = 0;
        key.medium_u = medium_u ? medium_u[i] : -1.0f;
        float t = 0; uint32_t prim = 0; int side = 0; bool hit;
        if (scope < 0) hit = closest_hit(S, r, tmin, tmax, key, t, prim, side);
        else { prim = (uint32_t)scope; hit = prim_t(S, prim, r.o, r.d, tmin, tmax, r.time, key, t, side); }
        rt_hit o; std::memset(&o, 0, sizeof(o));
        if (hit) {
            Hit h; prim_surface(S, prim, r.o, r.d, r.time, t, side, h);
            o.hit = 1; o.prim = prim; o.t = h.t; o.u = h.u; o.v = h.v;
            o.point[0] = h.point.x; o.point[1] = h.point.y; o.point[2] = h.point.z;
            o.normal[0] = h.normal.x; o.normal[1] = h.normal.y; o.normal[2] = h.normal.z;
            o.in_vec[0] = h.in_vec.x; o.in_vec[1] = h.in_vec.y; o.in_vec[2] = h.in_vec.z;
        }
        out[i] = o;
    }
}

void twin_camera_rays(const rt_camera* cam, uint32_t width, uint32_t height, uint64_t seed, uint64_t n, const uint32_t* px_i, const uint32_t* px_j,
                      const uint32_t* sample, rt_ray* out) {
    for (uint64_t k = 0; k < n; ++k) {
        Ray r = camera_ray(*cam, px_i[k], px_j[k], width, height, sample[k], (uint32_t)seed, (uint32_t)(seed >> 32));
        out[k].origin[0] = r.o.x; out[k].origin[1] = r.o.y; out[k].origin[2] = r.o.z;
        out[k].direction[0] = r.d.x; out[k].direction[1] = r.d.y; out[k].direction[2] = r.d.z; out[k].time = r.time;
    }
}

// One shading event on a given hit record: returns scattered flag, colour, new ray.
void twin_material_scatter(const rt_scene_desc* D, const rt_hit* h, uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t bounce, int32_t* scattered,
                           float* color, float* ray_o, float* ray_d) {
    const TwinScene scene(D, false); const SceneView& S = scene.V;
    Hit hh; hh.t = h->t; hh.u = h->u; hh.v = h->v; hh.prim = h->prim;
    hh.point = v3(h->point[0], h->point[1], h->point[2]); hh.normal = v3(h->normal[0], h->normal[1], h->normal[2]);
    hh.in_vec = v3(h->in_vec[0], h->in_vec[1], h->in_vec[2]);
    DrawKey key; key.k0 = (uint32_t)seed; key.k1 = (uint32_t)(seed >> 32); key.pixel = pixel; key.sample = sample; key.bounce = bounce; key.medium_u = -1.0f;
    Scatter sc = material_scatter(S, hh, key);
    *scattered = sc.scattered ? 1 : 0;
    color[0] = sc.color.x; color[1] = sc.color.y; color[2] = sc.color.z;
    ray_o[0] = sc.o.x; ray_o[1] = sc.o.y; ray_o[2] = sc.o.z; ray_d[0] = sc.d.x; ray_d[1] = sc.d.y; ray_d[2] = sc.d.z;
}

unsigned long long twin_generic_leaves(void) { return g_generic_leaves; }
void twin_set_sample_rays(const rt_ray* rays, uint64_t n) {
    g_sample_rays = SampleRays();
    for (uint64_t i = 0; i < n; ++i) g_sample_rays.add(rays[i].origin, rays[i].direction);
}
// Per-ray traversal work (nodes visited, leaves reached, primitive tests run) for analysis.
// trace (optional, n x trace_cap): the node-visit count at which each leaf was reached.
void twin_traverse_stats(const rt_scene_desc* D, const rt_ray* rays, uint64_t n, uint64_t seed, uint32_t* out /* n x 3 */, uint32_t* trace, uint32_t trace_cap) {
    const TwinScene scene(D); const SceneView& S = scene.V;
    for (uint64_t i = 0; i < n; ++i) {
        Ray r; r.o = v3(rays[i].origin[0], rays[i].origin[1], rays[i].origin[2]);
        r.d = v3(rays[i].direction[0], rays[i].direction[1], rays[i].direction[2]); r.time = rays[i].time;
        DrawKey key; key.k0 = (uint32_t)seed; key.k1 = (uint32_t)(seed >> 32); key.pixel = (uint32_t)i; key.sample = 0; key.bounce = 0; key.medium_u = -1.0f;
        g_tc = TwinCounters{0, 0, 0, trace ? trace + (size_t)i * trace_cap : nullptr, 0, trace_cap};
        float t = 0; uint32_t prim = 0; int side = 0;
        closest_hit(S, r, 0.001f, RT_INF_F, key, t, prim, side);
        out[3 * i] = (uint32_t)g_tc.nodes; out[3 * i + 1] = (uint32_t)g_tc.leaves; out[3 * i + 2] = (uint32_t)g_tc.prim_tests;
    }
}

// What-if analysis: node tests per ray if the reference's binary tree were walked nearer child first (NOT result-neutral
// for this crate; only to see what the fixed leaf order costs).  out: node tests per ray.
void twin_traverse_ordered(const rt_scene_desc* D, const rt_ray* rays, uint64_t n, uint32_t* out) {
    const TwinScene scene(D); const SceneView& S = scene.V;
    const rt_bvh_node* T = D->bvh;
    for (uint64_t i = 0; i < n; ++i) {
        Ray r; r.o = v3(rays[i].origin[0], rays[i].origin[1], rays[i].origin[2]);
        r.d = v3(rays[i].direction[0], rays[i].direction[1], rays[i].direction[2]); r.time = rays[i].time;
        DrawKey key; key.k0 = 1; key.k1 = 0; key.pixel = (uint32_t)i; key.sample = 0; key.bounce = 0; key.medium_u = -1.0f;
        const V3 inv = v3(1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z);
        float tmax = RT_INF_F; uint32_t nodes = 0;
        auto entry = [&](uint32_t k, float& tn) {
            ++nodes;
            const rt_bvh_node& N = T[k];
            float x0 = (N.min[0] - r.o.x) * inv.x, x1 = (N.max[0] - r.o.x) * inv.x, y0 = (N.min[1] - r.o.y) * inv.y, y1 = (N.max[1] - r.o.y) * inv.y;
            float z0 = (N.min[2] - r.o.z) * inv.z, z1 = (N.max[2] - r.o.z) * inv.z;
            tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), 0.001f));
            const float tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fminf(fmaxf(z0, z1), tmax));
            return tn <= tf;
        };
        uint32_t stack[128]; float stack_t[128]; int sp = 0;
        float t0;
        if (D->n_bvh_nodes && entry(0, t0)) { stack[sp] = 0; stack_t[sp++] = t0; }
        while (sp) {
            const uint32_t k = stack[--sp];
            if (stack_t[sp] > tmax) continue;
            if (T[k].prim != RT_BVH_NONE) {
                float t; int side;
                if (prim_t(S, T[k].prim & RT_NODE_PRIM_MASK, r.o, r.d, 0.001f, tmax, r.time, key, t, side)) tmax = t;
                continue;
            }
            const uint32_t a = k + 1, b = T[a].skip;
            float ta, tb;
            const bool ha = entry(a, ta), hb = b < T[k].skip && entry(b, tb);
            if (ha && hb) {
                if (ta <= tb) { stack[sp] = b; stack_t[sp++] = tb; stack[sp] = a; stack_t[sp++] = ta; }
                else { stack[sp] = a; stack_t[sp++] = ta; stack[sp] = b; stack_t[sp++] = tb; }
            } else if (ha) { stack[sp] = a; stack_t[sp++] = ta; }
            else if (hb) { stack[sp] = b; stack_t[sp++] = tb; }
        }
        out[i] = nodes;
    }
}

// The walk's node array as the library builds it from a scene description (rt::make_walk_nodes): returns the number of
// slots including the sentinel and copies up to `cap` of them (8 x uint32 per slot) to `out`.
uint32_t twin_walk_nodes(const rt_scene_desc* D, uint32_t* out, uint32_t cap) {
    std::vector<WalkNode> nodes;
    make_walk_nodes(D, nodes, g_sample_rays.size() ? &g_sample_rays : nullptr);
    const size_t n = std::min<size_t>(nodes.size(), cap);
    if (out && n) std::memcpy(out, nodes.data(), n * sizeof(WalkNode));
    return (uint32_t)nodes.size();
}

// What-if analysis: the same walk with up to `defer` leaves left pending (their primitive tests run, in order, only when
// the list is full or the tree is exhausted) while the walk goes on with the old upper bound.  out: node tests per ray.
void twin_traverse_deferred(const rt_scene_desc* D, const rt_ray* rays, uint64_t n, uint32_t defer, uint32_t* out) {
    const TwinScene scene(D); const SceneView& S = scene.V;
    for (uint64_t i = 0; i < n; ++i) {
        Ray r; r.o = v3(rays[i].origin[0], rays[i].origin[1], rays[i].origin[2]);
        r.d = v3(rays[i].direction[0], rays[i].direction[1], rays[i].direction[2]); r.time = rays[i].time;
        DrawKey key; key.k0 = 1; key.k1 = 0; key.pixel = (uint32_t)i; key.sample = 0; key.bounce = 0; key.medium_u = -1.0f;
        const V3 inv = v3(1.0f / r.d.x, 1.0f / r.d.y, 1.0f / r.d.z);
        const V3 oi = v3(r.o.x * inv.x, r.o.y * inv.y, r.o.z * inv.z);
        Walk w; walk_init(w, RT_INF_F);
        uint32_t pend_leaf[16], pend_node[16], n_pend = 0, nodes = 0;
        auto flush = [&]() {
            for (uint32_t k = 0; k < n_pend; ++k) {
                w.leaf_node = pend_node[k];
                if (!walk_leaf_fast(S, r, inv, 0.001f, pend_leaf[k], w)) walk_leaf_generic(S, r, 0.001f, key, pend_leaf[k], w);
            }
            n_pend = 0;
        };
        while (w.i < S.n_bvh_nodes * RT_NODE_STRIDE) {
            ++nodes;
            const uint32_t leaf = walk_step(S, inv, oi, 0.001f, w);
            if (leaf == RT_BVH_NONE) continue;
            pend_leaf[n_pend] = leaf; pend_node[n_pend] = w.leaf_node; ++n_pend;
            if (n_pend >= defer) flush();
        }
        flush();
        out[i] = nodes;
    }
}

// One camera path, bounce by bounce: out holds 12 floats per segment (origin, direction, t, prim or -1, throughput, value.x).
// Returns the number of segments written.
uint32_t twin_trace_sample(const rt_scene_desc* D, const rt_camera* cam, uint32_t width, uint32_t height, uint32_t i, uint32_t j, uint32_t sample, uint32_t depth,
                           uint64_t seed, float* out, uint32_t cap) {
    const TwinScene scene(D); const SceneView& S = scene.V;
    Ray r = camera_ray(*cam, i, j, width, height, sample, (uint32_t)seed, (uint32_t)(seed >> 32));
    V3 thr = v3(1, 1, 1), value = v3(0, 0, 0);
    DrawKey key; key.k0 = (uint32_t)seed; key.k1 = (uint32_t)(seed >> 32); key.pixel = j * width + i; key.sample = sample; key.medium_u = -1.0f;
    uint32_t n = 0;
    for (uint32_t b = 0; b < depth && n < cap; ++b) {
        key.bounce = b;
        float t = 0; uint32_t prim = 0; int side = 0;
        float* o = out + 12 * n; ++n;
        o[0] = r.o.x; o[1] = r.o.y; o[2] = r.o.z; o[3] = r.d.x; o[4] = r.d.y; o[5] = r.d.z;
        bool hit = closest_hit(S, r, 0.001f, RT_INF_F, key, t, prim, side);
        o[6] = t; o[7] = hit ? (float)prim : -1.0f;
        bool go = shade_step(S, r, thr, key, hit, t, prim, side, b + 1 >= depth, value);
        o[8] = thr.x; o[9] = thr.y; o[10] = thr.z; o[11] = value.x;
        if (!go) break;
    }
    return n;
}

float twin_perlin3(const uint8_t* perm256, float x, float y, float z) { return perlin3(perm256, v3(x, y, z)); }
// side_of_surface for n points: f64 points in, the fp32 origins a new ray gets out
void twin_side_of_surface(const double* points, const float* normal, int leaves_along_n, uint64_t n, float* out) {
    for (uint64_t i = 0; i < n; ++i) {
        const D3 p = d3(points[3 * i], points[3 * i + 1], points[3 * i + 2]);
        const V3 f = v3(p);
        const D3 e = p - d3(f);
        const uint32_t side = (e.x > 0.0 ? 1u : 0u) | (e.x < 0.0 ? 2u : 0u) | (e.y > 0.0 ? 4u : 0u) | (e.y < 0.0 ? 8u : 0u) | (e.z > 0.0 ? 16u : 0u) | (e.z < 0.0 ? 32u : 0u);
        const V3 r = side_of_surface(f, side, v3(normal[0], normal[1], normal[2]), leaves_along_n != 0);
        out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z;
    }
}
void twin_random_unit_vector(float u_phi, float u_z, float* out) { V3 r = random_unit_vector(u_phi, u_z); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
void twin_philox(const uint32_t* c, const uint32_t* k, uint32_t* out) {
    U4 r = philox4x32_10(c[0], c[1], c[2], c[3], k[0], k[1]); out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}
uint8_t twin_quantise(double x) { return quantise(x); }
// Pixels in the tiles of `rank` with a tile number below t_end (the place of tile t_end in the rank's pixel list).
uint64_t twin_owned_pixels_below(uint32_t W, uint32_t H, uint32_t tw, uint32_t th, uint32_t rank, uint32_t world, uint64_t t_end) {
    const TileGrid G{W, H, tw < W ? tw : W, th < H ? th : H, rank, world};
    return owned_pixels_below(G, t_end);
}

// The whole per-pixel loop (raygen -> closest_hit / shade_step per bounce -> f64 sum in sample order),
// i.e. what the kernels compute, on host threads.  mean_out: 3*w*h doubles, top row first.
void twin_render(const rt_scene_desc* D, const rt_camera* cam, uint32_t width, uint32_t height, uint64_t spp, uint64_t depth, uint64_t seed, int threads,
                 double* mean_out) {
    const TwinScene scene(D); const SceneView& S = scene.V;
    std::atomic<uint32_t> next_row(0);
    auto worker = [&]() {
            for (;;) {
            uint32_t row = next_row.fetch_add(1);
            if (row >= height) break;
            uint32_t j = height - 1 - row;
            for (uint32_t i = 0; i < width; ++i) {
                double acc[3] = {0, 0, 0};
                for (uint64_t s = 0; s < spp; ++s) {
                    Ray r = camera_ray(*cam, i, j, width, height, (uint32_t)s, (uint32_t)seed, (uint32_t)(seed >> 32));
                    V3 thr = v3(1, 1, 1), value = v3(0, 0, 0);
                    DrawKey key; key.k0 = (uint32_t)seed; key.k1 = (uint32_t)(seed >> 32); key.pixel = j * width + i; key.sample = (uint32_t)s; key.medium_u = -1.0f;
                    for (uint32_t b = 0; b < depth; ++b) {
                        key.bounce = b;
                        float t = 0; uint32_t prim = 0; int side = 0;
                        bool hit = closest_hit(S, r, 0.001f, RT_INF_F, key, t, prim, side);
                        if (!shade_step(S, r, thr, key, hit, t, prim, side, b + 1 >= depth, value)) break;
                    }
                    acc[0] += (double)value.x; acc[1] += (double)value.y; acc[2] += (double)value.z;
                }
                size_t o = 3 * ((size_t)row * width + i);
                for (int c = 0; c < 3; ++c) mean_out[o + c] = acc[c] / (double)spp;
            }
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; ++t) pool.emplace_back(worker);
    worker();
    for (auto& th : pool) th.join();
}

}  // extern "C"
