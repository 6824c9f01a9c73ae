// rt_hip.hip -- HIP kernels for gfx950 and the C ABI of include/rt_abi.h.
//
// Wavefront path tracer.  One pass traces `n_paths = owned_pixels * S` camera
// paths (S samples of every pixel this rank owns):
//
//   k_raygen    one thread per path: pixel() / Camera::ray (lib.rs:55-60, :389)
//   per bounce  k_traverse  closest hit through the BVH  (lib.rs:228-257 + shape.rs:248-426)
//               k_shade     Material::scatter + Texture::color_at (materials.rs:28, textures.rs:46),
//                           throughput update of ray_color (lib.rs:89-101), and compaction of the
//                           surviving rays into the next queue (wave ballot + per-block prefix,
//                           one atomic per block)
//   k_tail      once the queue is short: every remaining path runs to its end in one launch
//   k_resolve   per-pixel sum of the pass's sample values in sample order (lib.rs:65)
//   k_finalize  mean radiance -> full-frame f64 buffer (zeros for pixels of other ranks)
//
// Every random draw is Philox4x32-10 keyed on (pixel, sample, bounce, stream), each
// sample value is written to a slot of its own and summed in sample order, so the image
// does not depend on queue order, pass size, tile split or GPU count.
//
// HBM layout (per scene handle, sized for the largest pass): two ray queues of
// {float4 o.xyz|d.x, float4 d.yz|time|path, float4 throughput}, one hit queue of
// {t, prim|side} and one float4 sample value per path; per-pixel f64 sums.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "rt_core.h"

using namespace rt;

#define RT_BS 256
#define RT_HIT_NONE 0xFFFFFFFFu

// ------------------------------------------------------------------ device
// Queue traffic is streamed once and is far larger than the caches: it moves with non-temporal loads and stores, which
// leave the L2 to the scene's tables.
typedef float NtF4 __attribute__((ext_vector_type(4)));
typedef float NtF2 __attribute__((ext_vector_type(2)));
typedef uint32_t NtU2 __attribute__((ext_vector_type(2)));
struct RayQueue {
    float4* od0;  // o.x o.y o.z d.x
    float4* od1;  // d.y d.z time path(bits)
    float4* thr;  // throughput rgb, pixel(bits): the pixel's number j * width + i in the frame
};

struct PassParams {
    rt_camera cam;
    uint32_t width, height;
    uint32_t n_owned;      // pixels owned by this rank
    uint64_t owned_magic;  // floor(2^64 / n_owned) + 1 (0 when n_owned == 1): path / n_owned as a multiply-high
    uint32_t n_paths;      // n_owned * samples in this pass
    uint32_t sample_base;  // first sample index of the pass
    uint32_t k0, k1;       // seed
    uint32_t max_depth;
    uint32_t up_axis; float up_sign;  // the axis the camera's up vector (cam.v) is closest to, and its sign there
    uint32_t n_counts;     // queue lengths come in two arrays of this many entries, one behind the other (see queue_slot)
    uint32_t overlapped;   // the pass runs as two halves on two streams: kernels of one half run beside kernels of the other
    const uint32_t* owned; // global pixel id (j * width + i) of every owned pixel
};

// A queue has two ends.  Scattered rays that point up (along the axis the camera's up vector is closest to) are written
// from slot 0 upwards, the others from slot n_paths - 1 downwards, and a queue is read front part first: the rays a
// wave or a workgroup holds at a time then go to the same half of the scene, meet more of the same primitives and
// materials, and both the primitive tests of the walk and the shading diverge less (final_scene: shading 8 % faster,
// the walk 3 %; the kind of primitive a ray leaves is as good a key for the walk and half as good for shading).
// Entry i of a queue with n_front rays at the front:
__device__ __forceinline__ uint32_t queue_slot(uint32_t i, uint32_t n_front, uint32_t capacity) {
    return i < n_front ? i : capacity - 1u - (i - n_front);
}

// Path p of a pass is sample sample_base + p / n_owned of owned pixel p % n_owned.
__device__ __forceinline__ uint32_t path_sample(const PassParams& P, uint32_t path) {
    return P.sample_base + (P.owned_magic ? (uint32_t)__umul64hi((uint64_t)path, P.owned_magic) : path);
}
__device__ __forceinline__ DrawKey make_key(const PassParams& P, uint32_t path, uint32_t pixel, uint32_t bounce) {
    DrawKey k;
    k.k0 = P.k0; k.k1 = P.k1;
    k.pixel = pixel;
    k.sample = path_sample(P, path);
    k.bounce = bounce;
    k.medium_u = -1.0f;
    return k;
}

__global__ void __launch_bounds__(RT_BS) k_raygen(PassParams P, RayQueue q, uint32_t* counts, uint32_t n_counts) {
    uint32_t p = blockIdx.x * RT_BS + threadIdx.x;
    if (p == 0) {
        counts[0] = P.n_paths;  // camera rays: all at the front
        for (uint32_t i = 1; i < 2u * n_counts; ++i) counts[i] = 0;
    }
    if (p >= P.n_paths) return;
    const uint32_t sample = path_sample(P, p);
    const uint32_t pixel = P.owned[p - (sample - P.sample_base) * P.n_owned];
    Ray r = camera_ray(P.cam, pixel % P.width, pixel / P.width, P.width, P.height, sample, P.k0, P.k1);
    q.od0[p] = make_float4(r.o.x, r.o.y, r.o.z, r.d.x);
    q.od1[p] = make_float4(r.d.y, r.d.z, r.time, __uint_as_float(p));
    q.thr[p] = make_float4(1.0f, 1.0f, 1.0f, __uint_as_float(pixel));
}

// The queue lengths of a pass whose first bounce makes its own camera rays (k_shade<.., FIRST>): what k_raygen's first lane does.
__global__ void __launch_bounds__(64) k_reset_counts(uint32_t* counts, uint32_t n_counts, uint32_t n_paths) {
    for (uint32_t i = threadIdx.x; i < 2u * n_counts; i += 64u) counts[i] = i == 0 ? n_paths : 0u;
}

// A few rays of a queue, evenly spaced over it, for the host (see tune_tree): out[k] = {origin, direction} of entry
// (k + 1/2) * count / n; a zero direction where the queue is empty.
#define RT_TUNE_BOUNCES 16u      // bounces of the sampled pass that lend rays
#define RT_TUNE_PER_BOUNCE 20u   // rays per bounce
#define RT_TUNE_MIN_PATHS (1u << 18)  // a pass lends rays if it holds at least this many paths
__global__ void __launch_bounds__(64) k_sample_rays(RayQueue q, const uint32_t* count_ptr, uint32_t n_counts, uint32_t capacity, uint32_t n, float* out) {
    const uint32_t k = threadIdx.x;
    if (k >= n) return;
    const uint32_t n_front = count_ptr[0], count = n_front + count_ptr[n_counts];
    float* o = out + 6u * k;
    if (count == 0) { for (int c = 0; c < 6; ++c) o[c] = 0.0f; return; }
    const uint32_t i = (uint32_t)(((uint64_t)(2u * k + 1u) * count) / (2u * n));
    const uint32_t at = queue_slot(i < count ? i : count - 1u, n_front, capacity);
    const float4 a = q.od0[at], b = q.od1[at];
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y;
}

// The pixels this rank owns, tile by tile (TileGrid in rt_core.h), rows of a tile bottom-up and left to right:
// owned[k] = j * width + i.  Workgroup (x, y) writes rows [y * rows_per_block, (y + 1) * rows_per_block) of the rank's x-th
// tile; the tile's place in the list is the number of pixels in the rank's earlier tiles (owned_pixels_below: closed form
// but for one sum over the tile rows, which the workgroup's threads share).  (The order of the list only lays out the
// workspace; every draw is keyed by the pixel's number in the frame.)
#define RT_OWNED_BS 256
__global__ void __launch_bounds__(RT_OWNED_BS) k_owned_list(TileGrid G, uint32_t rows_per_block, uint32_t* owned) {
    __shared__ uint32_t last_column;
    const uint64_t tile = (uint64_t)G.rank + (uint64_t)blockIdx.x * G.world;
    if (threadIdx.x == 0) last_column = 0;
    __syncthreads();
    uint32_t mine = 0;
    for (uint32_t ty = threadIdx.x; ty < G.nty(); ty += RT_OWNED_BS) mine += owned_last_column_tile(G, ty, tile);
    if (mine) atomicAdd(&last_column, mine);
    __syncthreads();
    const uint64_t base = owned_pixels_below(G, tile, last_column);
    const uint32_t x0 = (uint32_t)(tile % G.ntx()) * G.tw, y0 = (uint32_t)(tile / G.ntx()) * G.th;
    const uint32_t cw = min(G.tw, G.W - x0), ch = min(G.th, G.H - y0);
    const uint32_t r0 = blockIdx.y * rows_per_block;
    if (r0 >= ch) return;
    const uint32_t rows = min(rows_per_block, ch - r0);
    for (uint64_t k = threadIdx.x; k < (uint64_t)rows * cw; k += RT_OWNED_BS) {
        const uint32_t y = r0 + (uint32_t)(k / cw), x = (uint32_t)(k % cw);
        owned[base + (uint64_t)y * cw + x] = (y0 + y) * G.W + x0 + x;
    }
}

// BVH nodes staged into LDS: each workgroup copies the node array once per launch and every ray of the
// launch walks it there.  The walk is a chain of dependent 32-byte node reads at a different address in
// every lane; from the vector L1 that costs a cache-line lookup per lane per step, from LDS a banked read.
// NODES_IN_LDS = false is the same kernel reading the nodes from global memory (trees too large for LDS).
#define RT_TRAV_BS 1024
#define RT_LDS_NODE_LIMIT (144u * 1024u)  // dynamic LDS the walk kernels may ask for
#define RT_LDS_BEHIND_NODES 16u           // k_traverse keeps its ray counter behind the node array
#define RT_LDS_SCENE_LIMIT (16u * 1024u)  // a scene whose tables (primitives, geometry, materials, textures, noise tables) fit in this goes to LDS whole
// A small scene's tables follow the node array into LDS (S.lds_scene_bytes != 0, set by the host when they fit): the general
// primitive test reads a primitive, its wrappers and its geometry one after the other, shading reads the material and the
// texture, Perlin noise makes 24 dependent look-ups in its permutation table -- chains of dependent loads that cost a trip
// to the L2 each from global memory and a fraction of that from LDS (the pointers in the SceneView are generic addresses;
// pointed into LDS the same code reads them with flat loads).
__device__ __forceinline__ uint32_t lds_align16(uint32_t bytes) { return (bytes + 15u) & ~15u; }
template <class T>
__device__ __forceinline__ const T* stage_table(const T* src, uint32_t count, unsigned char*& at) {
    const uint32_t words = (uint32_t)(sizeof(T) / 4u) * count;  // every table's element size is a multiple of four bytes
    static_assert(sizeof(T) % 4 == 0, "table element size");
    const uint32_t* s = reinterpret_cast<const uint32_t*>(src);
    uint32_t* d = reinterpret_cast<uint32_t*>(at);
    for (uint32_t k = threadIdx.x; k < words; k += blockDim.x) d[k] = s[k];
    const T* out = reinterpret_cast<const T*>(at);
    at += lds_align16(words * 4u);
    return count ? out : src;
}
// SIDES: the build of the kernel for scenes with hit points that need side_of_surface (rt_core.h); the other build has that
// code folded away (its mere presence cost the scenes without such surfaces 1 to 2 %: registers, spills).
// MULTI = false: the build for scenes without media flagged RT_PRIM_MULTI_SCATTER has that code folded away as well (k_traverse
// only: its register allocation is tight enough for the extra paths to cost spills in the walk itself).
template <bool NODES_IN_LDS, bool SIDES = false, bool MULTI = true>
__device__ __forceinline__ SceneView stage_nodes(const SceneView& S, uint4* lds) {
    SceneView V = S;
    V.side_rounding = SIDES ? 1u : 0u;
    if (!MULTI) V.multi_scatter = 0u;
    if (NODES_IN_LDS) {
        const uint4* src = reinterpret_cast<const uint4*>(S.bvh);
        for (uint32_t k = threadIdx.x; k < 2u * (S.n_bvh_nodes + 1u); k += blockDim.x) lds[k] = src[k];  // with the sentinel slot
        V.bvh = reinterpret_cast<const WalkNode*>(lds);
        if (S.lds_scene_bytes) {
            unsigned char* at = reinterpret_cast<unsigned char*>(lds + 2u * (S.n_bvh_nodes + 1u) + 1u);  // (one 16-byte slot behind the nodes is k_traverse's ray counter)
            V.prims = stage_table(S.prims, S.n_prims, at);
            V.spheres = stage_table(S.spheres, S.n_spheres, at);
            V.rects = stage_table(S.rects, S.n_rects, at);
            V.boxes = stage_table(S.boxes, S.n_boxes, at);
            V.xforms = stage_table(S.xforms, S.n_xforms, at);
            V.shade = stage_table(S.shade, S.n_prims, at);
            V.textures = stage_table(S.textures, S.n_textures, at);
            V.images = stage_table(S.images, S.n_images, at);
            V.perm = reinterpret_cast<const uint8_t*>(stage_table(reinterpret_cast<const uint32_t*>(S.perm), 64u * S.n_perm_tables, at));
        }
        __syncthreads();
    }
    return V;
}

// Closest hit for every ray of the queue: persistent waves, every lane walking its own ray.
//
// A lane has a ray or is idle; a lane with a ray steps over BVH nodes and notes the leaves it meets (up to two, see
// walk_step_flat), whose primitive tests are made later.  Each turn of the loop the wave either makes a run of node
// steps (branch-free, all lanes in lockstep) or, once enough lanes have a leaf pending or can no longer step, runs
// the oldest pending primitive test of every lane that has one; lanes whose ray is finished take the next ray of the
// workgroup's share of the queue (one LDS atomic per wave and turn).  The long primitive tests thus run for most of
// the wave at a time, and lanes stay busy although rays need very different numbers of steps (20 to 200 on
// final_scene).  The per-ray arithmetic is that of closest_hit() in rt_core.h.
#ifndef RT_LEAF_BATCH
#define RT_LEAF_BATCH 48      // run the primitive tests once this many of the 64 lanes have a leaf pending
#endif
#ifndef RT_REFILL_MIN
#define RT_REFILL_MIN 20      // fetch new rays once this many lanes are idle (a fetch waits on global memory)
#endif
#define RT_TRAV_GROUP 256u    // rays dealt to a workgroup at a time
#ifndef RT_STUCK_MAX
#define RT_STUCK_MAX 20       // ... or once this many lanes cannot step (two leaves pending, or out of the tree with one)
#endif
#ifndef RT_STEPS_PER_TURN
#define RT_STEPS_PER_TURN 10  // node steps between two looks at the wave's state
#endif
#ifdef RT_TRAV_STATS
// Analysis build only: what the waves of k_traverse spend their turns on (summed over all launches since the last read).
// 0 step turns, 1 lanes that could step at the start of a step turn, 2 leaf turns, 3 lanes with a leaf pending at a leaf turn,
// 4 / 5 / 6 lanes whose oldest leaf is a sphere / a box / something else at a leaf turn, 7 leaf turns that hold more than one kind,
// 8 refills, 9 rays taken, 10 / 11 leaves tried in the first / second round of the leaf turns, 12 / 13 leaf turns with a plain
// rectangle in the second round and the lanes that hold one, 14 / 15 the same for leaves that are neither sphere, box nor rectangle
__device__ unsigned long long g_trav_stats[16];
#define RT_STAT(k, v) do { if (lane == 0) st[k] += (v); } while (0)
#else
#define RT_STAT(k, v)
#endif
// GENERIC = false: no leaf of the tree takes the general primitive test (the host has looked); without that code the kernel needs
// 54 registers instead of its 80 with 8 spilled.
template <bool NODES_IN_LDS, bool MULTI = false, bool GENERIC = true>
__global__ void __launch_bounds__(NODES_IN_LDS ? RT_TRAV_BS : RT_BS, NODES_IN_LDS ? 6 : 1) k_traverse(SceneView S0, PassParams P, RayQueue q, const uint32_t* count_ptr,
                                                                                   uint2* hits, uint32_t bounce) {
    extern __shared__ uint4 lds_nodes[];
    // (the workgroup's ray counter stands behind the node array, so that the array starts at LDS address 0 and a node's
    // byte offset is its address)
    uint32_t& next_ray = *reinterpret_cast<uint32_t*>(lds_nodes + (NODES_IN_LDS ? 2u * (S0.n_bvh_nodes + 1u) : 0u));
    const SceneView S = stage_nodes<NODES_IN_LDS, false, MULTI>(S0, lds_nodes);
    const uint32_t n_front = count_ptr[0], count = n_front + count_ptr[P.n_counts];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n_nodes = S.n_bvh_nodes * RT_NODE_STRIDE;  // the sentinel's position: node positions are byte offsets
    const float tmin = 0.001f;  // lib.rs:90: (0.001, inf)
    // The queue is dealt out to the workgroups in groups of RT_TRAV_GROUP rays, round robin, so that every workgroup
    // gets an even sample of the queue whatever order it is in; inside a workgroup the waves draw from the workgroup's
    // rays through one LDS counter.  Local ray number L is ray ((L / G) * gridDim + blockIdx) * G + L % G of the queue.
    const uint32_t n_groups = (count + RT_TRAV_GROUP - 1) / RT_TRAV_GROUP;
    const uint32_t my_groups = n_groups > blockIdx.x ? (n_groups - 1 - blockIdx.x) / gridDim.x + 1 : 0;
    const uint32_t my_rays = my_groups * RT_TRAV_GROUP;
    if (threadIdx.x == 0) next_ray = 0;
    __syncthreads();

#ifdef RT_TRAV_STATS
    unsigned long long st[16] = {};
#endif
    bool active = false, exhausted = n_nodes == 0;
    Ray r; r.o = v3(0, 0, 0); r.d = v3(1, 1, 1); r.time = 0.0f;
    V3 inv = v3(1, 1, 1), ainv = v3(1, 1, 1), oi = v3(0, 0, 0);
    // The walk's state between turns: the next node, the upper end of the ray interval (the accepted hit's t once there
    // is one) and the accepted primitive with the box side in the top bits, RT_HIT_NONE while nothing is accepted.
    uint32_t node = n_nodes;  // a lane without a ray rests on the sentinel slot
    float t_end = RT_INF_F;
    uint32_t best = RT_HIT_NONE;
    uint32_t slot = 0, p0 = RT_BVH_NONE, p1 = RT_BVH_NONE;  // p0, p1: pending leaves (node indices), oldest first
    if (n_nodes == 0)  // nothing to hit: every ray misses
        for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) hits[queue_slot(i, n_front, P.n_paths)] = make_uint2(0u, RT_HIT_NONE);
    for (;;) {
        const unsigned long long idle = __ballot(!active);
        const uint32_t n_idle = (uint32_t)__popcll(idle);
        if (!exhausted && (n_idle >= RT_REFILL_MIN)) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&next_ray, n_idle);
            base = __shfl(base, 0);
            exhausted = base + n_idle >= my_rays;
            const uint32_t local = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
            const uint32_t mine = ((local / RT_TRAV_GROUP) * gridDim.x + blockIdx.x) * RT_TRAV_GROUP + local % RT_TRAV_GROUP;
#ifdef RT_TRAV_STATS
            { const uint32_t taken = (uint32_t)__popcll(__ballot(!active && local < my_rays && mine < count)); RT_STAT(8, 1); RT_STAT(9, taken); }
#endif
            if (!active && local < my_rays && mine < count) {
                const uint32_t at = queue_slot(mine, n_front, P.n_paths);
                const NtF4 a = __builtin_nontemporal_load(reinterpret_cast<const NtF4*>(&q.od0[at]));
                const NtF2 b = __builtin_nontemporal_load(reinterpret_cast<const NtF2*>(&q.od1[at]));
                r.o = v3(a.x, a.y, a.z); r.d = v3(a.w, b.x, b.y);
                inv = walk_cap(ray_inv(r.d));  // capped for the node tests; the box leaves get the infinities back
                ainv = abs3(inv);
                oi = v3(r.o.x * inv.x, r.o.y * inv.y, r.o.z * inv.z);
                node = has_negative_zero(r.d) ? n_nodes : 0u;  // (such a ray misses every bounding box: it is done at once)
                t_end = RT_INF_F; best = RT_HIT_NONE;
                slot = at; p0 = RT_BVH_NONE; p1 = RT_BVH_NONE;
                active = true;
            }
        }
        const unsigned long long act = __ballot(active);
        if (act == 0ull) {
            if (exhausted) break;
            continue;
        }
        // (every ballot is of one comparison, the rest is mask arithmetic on the scalar unit)
        const unsigned long long pending = act & __ballot(p0 != RT_BVH_NONE);
        const unsigned long long runnable = act & __ballot(p1 == RT_BVH_NONE) & __ballot(node < n_nodes);
        const unsigned long long stuck = act & ~runnable & pending;  // two leaves pending, or out of the tree with one
        if (runnable != 0ull && (uint32_t)__popcll(pending) < RT_LEAF_BATCH && (uint32_t)__popcll(stuck) < RT_STUCK_MAX) {
            RT_STAT(0, 1); RT_STAT(1, (uint32_t)__popcll(runnable));
            Walk w; w.i = node; w.tmax = fminf(t_end, RT_WALK_FAR);
            unsigned long long m0 = __ballot(p0 != RT_BVH_NONE), m1 = ~__ballot(p1 == RT_BVH_NONE);
#pragma unroll
            for (int k = 0; k < RT_STEPS_PER_TURN; ++k) walk_step_flat<NODES_IN_LDS>(S, inv, ainv, oi, tmin, w, p0, p1, m0, m1);
            node = w.i;
        } else {
#ifdef RT_TRAV_STATS
            {
                const uint32_t lf = (active && p0 != RT_BVH_NONE) ? node_at(S.bvh, p0).prim : 0u;
                const uint32_t ns = (uint32_t)__popcll(__ballot((lf & RT_NODE_PLAIN_SPHERE) != 0)), nb = (uint32_t)__popcll(__ballot((lf & RT_NODE_PLAIN_BOX) != 0));
                const uint32_t np = (uint32_t)__popcll(pending);
                RT_STAT(2, 1); RT_STAT(3, np); RT_STAT(4, ns); RT_STAT(5, nb); RT_STAT(6, np - ns - nb);
                RT_STAT(7, ((ns != 0) + (nb != 0) + (np - ns - nb != 0)) > 1 ? 1 : 0);
            }
#endif
            // Two rounds: first the lanes whose oldest leaf is a sphere, then those whose oldest leaf (by now) is anything
            // else.  A wave runs both kinds of test in nearly every turn anyway (its lanes hold both kinds); this way a lane
            // with a sphere and then a box pending has both tried in one turn, in their order.
#pragma unroll
            for (int round = 0; round < 2; ++round) {
                const bool has = active && p0 != RT_BVH_NONE;
                const uint32_t leaf = has ? node_at(S.bvh, p0).prim : 0u;
#ifdef RT_TRAV_STATS
                { const uint32_t served = (uint32_t)__popcll(__ballot(has && ((leaf & RT_NODE_PLAIN_SPHERE) != 0u) == (round == 0))); RT_STAT(10 + round, served); }
                if (round == 1) {
                    const bool mine = has && (leaf & RT_NODE_PLAIN_SPHERE) == 0u;
                    const uint32_t n_rect = (uint32_t)__popcll(__ballot(mine && (leaf & RT_NODE_PLAIN_RECT) != 0u));
                    const uint32_t n_other = (uint32_t)__popcll(__ballot(mine && (leaf & (RT_NODE_PLAIN_RECT | RT_NODE_PLAIN_BOX)) == 0u));
                    RT_STAT(12, n_rect != 0 ? 1 : 0); RT_STAT(13, n_rect); RT_STAT(14, n_other != 0 ? 1 : 0); RT_STAT(15, n_other);
                }
#endif
                if (has && ((leaf & RT_NODE_PLAIN_SPHERE) != 0u) == (round == 0)) {
                    Walk w; walk_init(w, t_end);
                    w.leaf_node = p0;
                    // (the ray's time is not loaded with the ray: only a moving sphere needs it)
                    if (round == 0 && (leaf & RT_NODE_PLAIN_RECT) != 0u) r.time = q.od1[slot].z;
                    if (!walk_leaf_fast(S, r, inv, tmin, leaf, w) && GENERIC) {  // these leaves also need the path and pixel numbers (they key a medium's draw)
                        const float4 b = q.od1[slot];
                        r.time = b.z;
                        walk_leaf_generic(S, r, tmin, make_key(P, __float_as_uint(b.w), __float_as_uint(q.thr[slot].w), bounce), leaf, w);
                    }
                    if (w.found) { t_end = w.t; best = w.prim | ((uint32_t)w.side << 29); }
                    p0 = p1; p1 = RT_BVH_NONE;
                }
            }
        }
        if (active && p0 == RT_BVH_NONE && node >= n_nodes) {
            hits[slot] = make_uint2(best != RT_HIT_NONE ? __float_as_uint(t_end) : 0u, best);  // (a cached store: the lanes of a wave finish at different times, the L2 gathers their 8 bytes into lines)
            active = false;
        }
    }
#ifdef RT_TRAV_STATS
    if (lane == 0) for (int k = 0; k < 16; ++k) if (st[k]) atomicAdd(&g_trav_stats[k], st[k]);
#endif
}

#ifdef RT_SHADE_STATS
// Analysis build only: what the waves of k_shade (hit records from k_traverse) hold, summed over all launches since the last
// read.  Pairs (waves with at least one such lane, lanes): 0 all passes / live lanes, 1 hits, 2 misses, 3 spheres, 4 boxes,
// 5 rectangles, 6 media, 7 behind wrappers, 8 surface coordinates wanted, 9 Lambertian with an inlined solid colour, 10 Lambertian
// with a texture call, 11 metal, 12 dielectric, 13 light, 14 isotropic, 15 checker, 16 noise, 17 image texture, 18 moving spheres
__device__ unsigned long long g_shade_stats[40];
#endif
// Workgroup size of k_shade.  Each workgroup pass ends in one atomic on the length of the next queue; at 256 threads
// the rate of those atomics (all on one address) bounds the kernel, at 512 it runs within a few per cent of the same
// kernel without compaction, which moves its ~100 bytes per segment at over half of the HBM peak.
// Register budgets: k_traverse is held to 80 VGPRs (4 waves per SIMD = 320 of the 512 registers of a lane; its build without the
// general primitive test takes 52) and k_shade to 96, so that one shading workgroup (2 waves per SIMD) fits beside the traversal workgroup of the pass's other half.
#define RT_SHADE_BS 512
#ifndef RT_SHADE_PRIO
#define RT_SHADE_PRIO 3
#endif
#ifndef RT_SCAN_TWO
#define RT_SCAN_TWO 1  // the first launch of a scanned pass runs two bounces in scenes without long-test leaves
#endif
#ifndef RT_SCAN_FIRST
#define RT_SCAN_FIRST 1
#endif
#define RT_SCAN_MAX_LEAVES 16u   // leaves up to which a bounce runs as one kernel that scans them
// How k_shade gets a ray's closest hit:
//   SHADE_HITS  from the hit records of a k_traverse launch;
//   SHADE_WALK  by walking the tree itself (closest_hit() of rt_core.h, every lane its own walk, node array staged into LDS);
//   SHADE_SCAN  by trying the leaves one after the other, all lanes of the wave at the same leaf (closest_hit_scan below).
// For trees of a handful of leaves (the Cornell rooms, the perlin scene) a walk is a few steps, and one launch per bounce
// that reads a ray once and writes its successor moves 96 bytes per segment where the two-kernel form moves 147 and pays
// two launches.  Same arithmetic, same bytes in the image.
enum { SHADE_HITS = 0, SHADE_WALK = 1, SHADE_SCAN = 2 };

// The closest hit for a tree of a few leaves: the wave goes through the leaves in their order, every lane testing the same
// leaf at the same time.  That is BvhNode::scatter_ (lib.rs:228-257) without the inner nodes, which only cull: a leaf is tried
// iff the ray meets the leaf's own box within (tmin, current tmax), and a ray that does has met every box around it (they
// contain it, and the slab test is monotone in the bounds), so the same leaves are tried in the same order against the same
// bounds as in closest_hit().  The kind of the leaf is the same for all lanes, so its test is straight code for the wave
// instead of each lane's own branch, and a scene of eight objects costs eight box tests and the tests of the leaves some lane
// enters, not a persistent kernel's turns.  `live`: lanes without a ray go through the motions with every test masked off.
template <bool GENERIC>
__device__ __forceinline__ bool closest_hit_scan(const SceneView& S, const Ray& r, bool live, float tmin, const DrawKey& key, float& best_t, uint32_t& best_prim,
                                                 int& best_side) {
    const bool ok = live && !has_negative_zero(r.d);
    const V3 inv_c = walk_cap(ray_inv(r.d));
    const V3 oi = v3(r.o.x * inv_c.x, r.o.y * inv_c.y, r.o.z * inv_c.z);
    Walk w; walk_init(w, RT_INF_F);
    const uint32_t n = S.n_bvh_nodes * RT_NODE_STRIDE;
    for (uint32_t pos = 0; pos < n;) {  // (the same position in every lane)
        const WalkNode& N = node_at(S.bvh, pos);
        const uint32_t leaf = __builtin_amdgcn_readfirstlane(N.prim);
        if (leaf < RT_NODE_LEAF) { pos += RT_NODE_STRIDE; continue; }  // an inner node: on to its first child
        const uint32_t skip = __builtin_amdgcn_readfirstlane(N.skip);
        const bool pass = ok && node_hit(N, inv_c, oi, tmin, fminf(w.tmax, RT_WALK_FAR));
        if (__ballot(pass) != 0ull && pass) {
            w.leaf_node = pos;
            if (!walk_leaf_fast(S, r, inv_c, tmin, leaf, w) && GENERIC) walk_leaf_generic(S, r, tmin, key, leaf, w);
        }
        pos = skip;  // a leaf's skip link also jumps over its payload slots
    }
    best_t = w.t; best_prim = w.prim; best_side = w.side;
    return w.found;
}

// Which end of the next queue a scattered ray of a scanned scene goes to.  The leaves whose test is long (anything but a
// plain sphere, rectangle or box: shapes behind wrappers, media -- the f64 path of prim_t) are tried by a wave iff one of its
// lanes meets the leaf's box, so a wave of rays that all pass such leaves by skips their tests altogether: rays that meet the
// box of at least one of them go to the front of the queue, the others to the back, and a wave of the next bounce holds
// (but for the one at the boundary) rays of one kind.  The test is the scan's own box test with the interval still
// unbounded above, so "passes by" is certain; it costs a reciprocal direction and one slab test per such leaf.  Scenes
// without such leaves run the build of the kernel without this code (LONG_LEAVES = false: the key of the general case, up
// or down along the camera's up axis; the code's mere presence cost perlin 6 %).  A queue's order never changes the image.
__device__ __forceinline__ bool scan_queue_end(const SceneView& S, const Ray& r) {
    const V3 inv_c = walk_cap(ray_inv(r.d));
    const V3 oi = v3(r.o.x * inv_c.x, r.o.y * inv_c.y, r.o.z * inv_c.z);
    bool meets = false;
    const uint32_t n = S.n_bvh_nodes * RT_NODE_STRIDE;
    for (uint32_t pos = 0; pos < n;) {  // (the same position in every lane)
        const WalkNode& N = node_at(S.bvh, pos);
        const uint32_t leaf = __builtin_amdgcn_readfirstlane(N.prim);
        if (leaf < RT_NODE_LEAF) { pos += RT_NODE_STRIDE; continue; }
        if (!(leaf & (RT_NODE_PLAIN_SPHERE | RT_NODE_PLAIN_BOX | RT_NODE_PLAIN_RECT))) meets = meets || node_hit(N, inv_c, oi, 0.001f, RT_WALK_FAR);
        pos = __builtin_amdgcn_readfirstlane(N.skip);
    }
    return meets && !has_negative_zero(r.d);
}

// A scanned bounce (SHADE_SCAN) runs in workgroups of 256: its waves do a whole segment's work between two barriers and a CU
// holds sixteen of them at the kernel's 128 registers, so four workgroups per CU are in four different phases (load, scan,
// shade, store) where two of 512 are in two (cornell +2 %, cornell_smoke +7 %, perlin unchanged; its rate of atomics,
// 58 M/s per queue end on cornell, stays below what one address takes; 128 threads: 15 to 30 % slower).
#ifndef RT_SCAN_GRID_PER_CU
#define RT_SCAN_GRID_PER_CU 8u  // workgroups of a scanned bounce per CU (four are resident at a time)
#endif
#ifndef RT_TAIL_PER_CU
#define RT_TAIL_PER_CU 512u  // rays per CU at which the queue is short enough for k_tail to take the rest of every path
#endif
#ifndef RT_SCAN_SPLIT
#define RT_SCAN_SPLIT 1  // scanned bounces of a queue ordered by long-test leaves run as two launches (front, back)
#endif
#ifndef RT_PLAIN_SCAN_GRID_PER_CU
#define RT_PLAIN_SCAN_GRID_PER_CU 15u  // ... of the builds without long-test leaves (five resident)
#endif
__host__ __device__ constexpr uint32_t shade_bs(int mode) { return mode == 2 /* SHADE_SCAN */ ? 256u : RT_SHADE_BS; }
#ifndef RT_TWO_WAVES
#define RT_TWO_WAVES 5
#endif
#ifndef RT_PLAIN_SCAN_WAVES
#define RT_PLAIN_SCAN_WAVES 5
#endif
#ifndef RT_FRONT_SCAN_WAVES
#define RT_FRONT_SCAN_WAVES 5
#endif
#ifndef RT_LONG_FIRST_WAVES
#define RT_LONG_FIRST_WAVES 4
#endif
#ifndef RT_BACK_SCAN_WAVES
#define RT_BACK_SCAN_WAVES 5
#endif
// PART (scanned bounces of scenes with long-test leaves, whose queue is ordered by them): 0 = the whole queue, 1 = its front part,
// 2 = its back part, whose rays pass every long-test leaf by (scan_queue_end said so with the scan's own box test), in a build
// without the general primitive test.
template <int MODE, bool SIDES, bool LONG_LEAVES = false, bool FIRST = false, bool TWO = false, int PART = 0>
__global__ void __launch_bounds__(shade_bs(MODE), MODE != SHADE_HITS ? (MODE == SHADE_SCAN && !LONG_LEAVES ? (TWO ? RT_TWO_WAVES : RT_PLAIN_SCAN_WAVES) : (PART == 2 ? RT_BACK_SCAN_WAVES : (PART == 1 ? RT_FRONT_SCAN_WAVES : (MODE == SHADE_SCAN && FIRST ? RT_LONG_FIRST_WAVES : 4)))) : 5) k_shade(SceneView S0, PassParams P, RayQueue qin, const uint2* hits, RayQueue qout, const uint32_t* count_ptr,
                                                 uint32_t* next_count, float4* values, uint32_t bounce) {
    constexpr uint32_t BS = shade_bs(MODE);
    extern __shared__ uint4 lds_nodes[];
    __shared__ uint32_t wave_cnt[2][BS / 64];
    __shared__ uint32_t block_base[2];
    // Shading waves that run beside the traversal workgroup of the pass's other half (two streams) spend most of their time
    // waiting on queue memory: at a higher issue priority their few instructions go ahead of the traversal's whenever they
    // are ready, the shading kernel is gone long before the traversal next to it ends, and the frame is left with the sum
    // of the two kernels' instruction issue (final_scene: 507 -> 573 Msamples/s; raising the traversal's priority instead
    // changed nothing).  On its own the kernel is a few per cent slower with the priority raised, so it is only raised there.
    if (MODE == SHADE_HITS && P.overlapped) __builtin_amdgcn_s_setprio(RT_SHADE_PRIO);
    const SceneView S = stage_nodes<MODE != SHADE_HITS, SIDES>(S0, lds_nodes);
    // FIRST (scanned scenes): bounce 0 makes its camera rays itself instead of reading them from a queue that k_raygen has
    // written (96 bytes per path less through memory; perlin's paths are 2.3 segments long)
    const uint32_t n_front = FIRST ? P.n_paths : count_ptr[0], count = FIRST ? P.n_paths : n_front + count_ptr[P.n_counts];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t lo = PART == 2 ? n_front : 0u, hi = PART == 1 ? n_front : count;
    for (uint32_t base = lo + blockIdx.x * BS; base < hi; base += gridDim.x * BS) {
        const uint32_t i = base + threadIdx.x;
        bool alive = false, to_front = false;
        Ray r; V3 thr; uint32_t path = 0, pixel = 0;
        if (MODE == SHADE_SCAN) {
            // the scan is the wave's, so the hit is found outside the per-lane branch below
            DrawKey key = make_key(P, 0u, 0u, bounce);
            r.o = v3(0.0f, 0.0f, 0.0f); r.d = v3(1.0f, 1.0f, 1.0f); r.time = 0.0f; thr = v3(0.0f, 0.0f, 0.0f);
            const bool live = i < hi;
            if (live && FIRST) {  // (as k_raygen)
                path = i;
                const uint32_t sample = path_sample(P, path);
                pixel = P.owned[path - (sample - P.sample_base) * P.n_owned];
                r = camera_ray(P.cam, pixel % P.width, pixel / P.width, P.width, P.height, sample, P.k0, P.k1);
                thr = v3(1.0f, 1.0f, 1.0f);
                key = make_key(P, path, pixel, bounce);
            } else if (live) {
                const uint32_t at = queue_slot(i, n_front, P.n_paths);
                const NtF4 a = __builtin_nontemporal_load(reinterpret_cast<const NtF4*>(&qin.od0[at])), b = __builtin_nontemporal_load(reinterpret_cast<const NtF4*>(&qin.od1[at])),
                           c = __builtin_nontemporal_load(reinterpret_cast<const NtF4*>(&qin.thr[at]));
                r.o = v3(a.x, a.y, a.z); r.d = v3(a.w, b.x, b.y); r.time = b.z;
                path = __float_as_uint(b.w);
                thr = v3(c.x, c.y, c.z);
                pixel = __float_as_uint(c.w);
                key = make_key(P, path, pixel, bounce);
            }
            // TWO: the lanes run this bounce and the next before anything goes to the queue (a scattered ray stays in its lane;
            // the caller's next launch is for bounce + 2 and reads the counters two places on).  Draws are keyed by (pixel,
            // sample, bounce), so who runs a bounce changes no byte.
            alive = live;
#pragma nounroll
            for (uint32_t k = 0; k < (TWO ? 2u : 1u); ++k) {
                key.bounce = bounce + k;
                float t = 0.0f; uint32_t prim = 0; int side = 0;
                const bool hit = closest_hit_scan<LONG_LEAVES && PART != 2>(S, r, alive, 0.001f, key, t, prim, side);  // lib.rs:90: (0.001, inf); without long-test leaves no leaf takes the general test
                if (alive) {
                    V3 value;
                    alive = shade_step(S, r, thr, key, hit, t, prim, side, bounce + k + 1 >= P.max_depth, value);
                    if (!alive) __builtin_nontemporal_store(NtF4{value.x, value.y, value.z, 0.0f}, reinterpret_cast<NtF4*>(&values[path]));
                }
            }
            if (alive && !LONG_LEAVES) to_front = (P.up_axis == 1 ? r.d.y : (P.up_axis == 0 ? r.d.x : r.d.z)) * P.up_sign > 0.0f;
            if (LONG_LEAVES) to_front = alive && scan_queue_end(S, r);
        } else if (i < hi) {
            const uint32_t at = queue_slot(i, n_front, P.n_paths);
            const NtF4 a_ = __builtin_nontemporal_load(reinterpret_cast<const NtF4*>(&qin.od0[at])), b_ = __builtin_nontemporal_load(reinterpret_cast<const NtF4*>(&qin.od1[at])),
                       c_ = __builtin_nontemporal_load(reinterpret_cast<const NtF4*>(&qin.thr[at]));
            float4 a = make_float4(a_.x, a_.y, a_.z, a_.w), b = make_float4(b_.x, b_.y, b_.z, b_.w), c = make_float4(c_.x, c_.y, c_.z, c_.w);
            r.o = v3(a.x, a.y, a.z); r.d = v3(a.w, b.x, b.y); r.time = b.z;
            path = __float_as_uint(b.w);
            thr = v3(c.x, c.y, c.z);
            pixel = __float_as_uint(c.w);
            DrawKey key = make_key(P, path, pixel, bounce);
            V3 value;
            bool hit; float t = 0.0f; uint32_t prim = 0; int side = 0;
            if (MODE == SHADE_WALK) {
                hit = closest_hit(S, r, 0.001f, RT_INF_F, key, t, prim, side);  // lib.rs:90: (0.001, inf)
            } else {
                const NtU2 h_ = __builtin_nontemporal_load(reinterpret_cast<const NtU2*>(&hits[at]));
                hit = h_.y != RT_HIT_NONE; t = __uint_as_float(h_.x); prim = h_.y & 0x1FFFFFFFu; side = (int)(h_.y >> 29);
            }
#ifdef RT_SHADE_STATS
            if (MODE == SHADE_HITS) {
                const ShadeRec R = S.shade[hit ? prim : 0u];
                const uint32_t tk = (hit && (R.mat.kind == RT_MAT_LAMBERTIAN || R.mat.kind == RT_MAT_DIFFUSE_LIGHT || R.mat.kind == RT_MAT_ISOTROPIC) && !(R.mat.reserved[0] & RT_MAT_SOLID_INLINE))
                                        ? S.textures[R.mat.texture].kind : 0xFFu;
                const bool sphere = R.prim.geom <= RT_GEOM_MOVING_SPHERE, medium = (R.prim.flags & RT_PRIM_MEDIUM) != 0;
                const bool c[19] = {true, hit, !hit, hit && sphere && !medium, hit && R.prim.geom == RT_GEOM_BOX && !medium, hit && R.prim.geom == RT_GEOM_RECT && !medium, hit && medium,
                                    hit && R.prim.xf_count > 0, hit && !(R.mat.reserved[0] & RT_MAT_SKIP_UV), hit && R.mat.kind == RT_MAT_LAMBERTIAN && tk == 0xFFu,
                                    hit && R.mat.kind == RT_MAT_LAMBERTIAN && tk != 0xFFu, hit && R.mat.kind == RT_MAT_METAL, hit && R.mat.kind == RT_MAT_DIELECTRIC,
                                    hit && R.mat.kind == RT_MAT_DIFFUSE_LIGHT, hit && R.mat.kind == RT_MAT_ISOTROPIC, tk == RT_TEX_CHECKERED, tk == RT_TEX_NOISE, tk == RT_TEX_IMAGE,
                                    hit && R.prim.geom == RT_GEOM_MOVING_SPHERE};
                for (int k = 0; k < 19; ++k) {
                    const unsigned long long m = __ballot(c[k]);
                    if (m != 0ull && __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)) == 0u && c[k]) {
                        atomicAdd(&g_shade_stats[2 * k], 1ull); atomicAdd(&g_shade_stats[2 * k + 1], (unsigned long long)__popcll(m));
                    }
                }
            }
#endif
            alive = shade_step(S, r, thr, key, hit, t, prim, side, bounce + 1 >= P.max_depth, value);
            if (!alive) __builtin_nontemporal_store(NtF4{value.x, value.y, value.z, 0.0f}, reinterpret_cast<NtF4*>(&values[path]));
            else to_front = (P.up_axis == 1 ? r.d.y : (P.up_axis == 0 ? r.d.x : r.d.z)) * P.up_sign > 0.0f;
        }
        // compaction, per end of the queue: ballot + lane prefix inside the wave, one atomic per block and end
        const unsigned long long mask_f = __ballot(alive && to_front), mask_b = __ballot(alive && !to_front);
        const unsigned long long mask = to_front ? mask_f : mask_b;
        const uint32_t prefix = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
        if (lane == 0) { wave_cnt[0][wave] = (uint32_t)__popcll(mask_f); wave_cnt[1][wave] = (uint32_t)__popcll(mask_b); }
        __syncthreads();
        if (threadIdx.x < 2) {
            const uint32_t e = threadIdx.x;
            uint32_t total = 0;
            for (int w = 0; w < (int)(BS / 64); ++w) { uint32_t c = wave_cnt[e][w]; wave_cnt[e][w] = total; total += c; }
            block_base[e] = total ? atomicAdd(next_count + e * P.n_counts, total) : 0u;
        }
        __syncthreads();
        if (alive) {
            const uint32_t k = block_base[to_front ? 0 : 1] + wave_cnt[to_front ? 0 : 1][wave] + prefix;
            const uint32_t slot = to_front ? k : P.n_paths - 1u - k;
            __builtin_nontemporal_store(NtF4{r.o.x, r.o.y, r.o.z, r.d.x}, reinterpret_cast<NtF4*>(&qout.od0[slot]));
            __builtin_nontemporal_store(NtF4{r.d.y, r.d.z, r.time, __uint_as_float(path)}, reinterpret_cast<NtF4*>(&qout.od1[slot]));
            __builtin_nontemporal_store(NtF4{thr.x, thr.y, thr.z, __uint_as_float(pixel)}, reinterpret_cast<NtF4*>(&qout.thr[slot]));
        }
        __syncthreads();
    }
}

// Every ray left in the queue runs the remaining iterations of ray_color in one launch.
template <bool NODES_IN_LDS, bool SIDES, bool GENERIC = true>
__global__ void __launch_bounds__(NODES_IN_LDS ? RT_TRAV_BS : RT_BS) k_tail(SceneView S0, PassParams P, RayQueue q, const uint32_t* count_ptr, float4* values, uint32_t* segment_counter,
                                                     uint32_t first_bounce) {
    extern __shared__ uint4 lds_nodes[];
    const SceneView S = stage_nodes<NODES_IN_LDS, SIDES>(S0, lds_nodes);
    const uint32_t n_front = count_ptr[0], count = n_front + count_ptr[P.n_counts];
    uint32_t segments = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
        const uint32_t at = queue_slot(i, n_front, P.n_paths);
        float4 a = q.od0[at], b = q.od1[at], c = q.thr[at];
        Ray r; r.o = v3(a.x, a.y, a.z); r.d = v3(a.w, b.x, b.y); r.time = b.z;
        const uint32_t path = __float_as_uint(b.w);
        V3 thr = v3(c.x, c.y, c.z);
        V3 value = v3(0.0f, 0.0f, 0.0f);
        DrawKey key = make_key(P, path, __float_as_uint(c.w), first_bounce);
        for (uint32_t bounce = first_bounce; bounce < P.max_depth; ++bounce) {
            key.bounce = bounce;
            ++segments;
            float t = 0.0f; uint32_t prim = 0; int side = 0;
            bool hit = closest_hit<GENERIC>(S, r, 0.001f, RT_INF_F, key, t, prim, side);
            if (!shade_step(S, r, thr, key, hit, t, prim, side, bounce + 1 >= P.max_depth, value)) break;
        }
        values[path] = make_float4(value.x, value.y, value.z, 0.0f);
    }
    // segments traced by this launch (profiling only): one atomic per wave
    for (int off = 32; off > 0; off >>= 1) segments += __shfl_down(segments, off);
    if ((threadIdx.x & 63u) == 0 && segments) atomicAdd(segment_counter, segments);
}

// accum[pixel] += value(sample s) for s ascending: the f64 sum of lib.rs:65 in sample order.
__global__ void __launch_bounds__(RT_BS) k_resolve(const float4* values, double* accum, uint32_t n_owned, uint32_t n_samples) {
    uint32_t pl = blockIdx.x * RT_BS + threadIdx.x;
    if (pl >= n_owned) return;
    double r = accum[3 * (size_t)pl], g = accum[3 * (size_t)pl + 1], b = accum[3 * (size_t)pl + 2];
    for (uint32_t s = 0; s < n_samples; ++s) {
        float4 v = values[(size_t)s * n_owned + pl];
        r += (double)v.x; g += (double)v.y; b += (double)v.z;
    }
    accum[3 * (size_t)pl] = r; accum[3 * (size_t)pl + 1] = g; accum[3 * (size_t)pl + 2] = b;
}

// mean = sum / spp (lib.rs:66) scattered to the output frame: row 0 = top = j = height-1 (lib.rs:21-23).
__global__ void __launch_bounds__(RT_BS) k_finalize(const double* accum, const uint32_t* owned, uint32_t n_owned, uint32_t width, uint32_t height,
                                                    double spp, double* mean_out) {
    uint32_t pl = blockIdx.x * RT_BS + threadIdx.x;
    if (pl >= n_owned) return;
    uint32_t pixel = owned[pl], i = pixel % width, j = pixel / width;
    size_t o = 3 * ((size_t)(height - 1 - j) * width + i);
    mean_out[o] = accum[3 * (size_t)pl] / spp;
    mean_out[o + 1] = accum[3 * (size_t)pl + 1] / spp;
    mean_out[o + 2] = accum[3 * (size_t)pl + 2] / spp;
}

__global__ void __launch_bounds__(RT_BS) k_quantise(const double* mean, uint64_t n, uint8_t* out) {
    uint64_t i = (uint64_t)blockIdx.x * RT_BS + threadIdx.x;
    if (i < n) out[i] = quantise(mean[i]);
}

__global__ void __launch_bounds__(RT_BS) k_intersect_batch(SceneView S, int64_t scope, const rt_ray* rays, uint64_t n, float tmin, float tmax,
                                                           uint32_t k0, uint32_t k1, const float* medium_u, rt_hit* out) {
    uint64_t i = (uint64_t)blockIdx.x * RT_BS + threadIdx.x;
    if (i >= n) return;
    rt_ray in = rays[i];
    Ray r; r.o = v3(in.origin[0], in.origin[1], in.origin[2]); r.d = v3(in.direction[0], in.direction[1], in.direction[2]); r.time = in.time;
    DrawKey key; key.k0 = k0; key.k1 = k1; key.pixel = (uint32_t)i; key.sample = 0; key.bounce = 0;
    key.medium_u = medium_u ? medium_u[i] : -1.0f;
    float t = 0.0f; uint32_t prim = 0; int side = 0;
    bool hit;
    if (scope < 0) hit = closest_hit(S, r, tmin, tmax, key, t, prim, side);
    else { prim = (uint32_t)scope; hit = prim_t(S, prim, r.o, r.d, tmin, tmax, r.time, key, t, side); }
    rt_hit o;
    memset(&o, 0, sizeof(o));
    if (hit) {
        Hit h;
        prim_surface(S, prim, r.o, r.d, r.time, t, side, h);
        o.hit = 1; o.prim = prim; o.t = h.t; o.u = h.u; o.v = h.v;
        o.point[0] = h.point.x; o.point[1] = h.point.y; o.point[2] = h.point.z;
        o.normal[0] = h.normal.x; o.normal[1] = h.normal.y; o.normal[2] = h.normal.z;
        o.in_vec[0] = h.in_vec.x; o.in_vec[1] = h.in_vec.y; o.in_vec[2] = h.in_vec.z;
    }
    out[i] = o;
}

__global__ void __launch_bounds__(RT_BS) k_scatter_batch(SceneView S, const rt_hit* hits, uint64_t n, uint32_t k0, uint32_t k1, const uint32_t* keys, rt_scatter* out) {
    uint64_t i = (uint64_t)blockIdx.x * RT_BS + threadIdx.x;
    if (i >= n) return;
    const rt_hit in = hits[i];
    Hit h; h.t = in.t; h.u = in.u; h.v = in.v; h.prim = in.prim;
    h.point = v3(in.point[0], in.point[1], in.point[2]); h.normal = v3(in.normal[0], in.normal[1], in.normal[2]);
    h.in_vec = v3(in.in_vec[0], in.in_vec[1], in.in_vec[2]);
    DrawKey key; key.k0 = k0; key.k1 = k1; key.medium_u = -1.0f;
    key.pixel = keys ? keys[3 * i] : (uint32_t)i; key.sample = keys ? keys[3 * i + 1] : 0u; key.bounce = keys ? keys[3 * i + 2] : 0u;
    const Scatter sc = material_scatter(S, h, key);
    rt_scatter o;
    o.scattered = sc.scattered ? 1 : 0;
    o.color[0] = sc.color.x; o.color[1] = sc.color.y; o.color[2] = sc.color.z;
    o.origin[0] = sc.o.x; o.origin[1] = sc.o.y; o.origin[2] = sc.o.z;
    o.direction[0] = sc.d.x; o.direction[1] = sc.d.y; o.direction[2] = sc.d.z;
    out[i] = o;
}

__global__ void __launch_bounds__(RT_BS) k_camera_rays(rt_camera cam, uint32_t width, uint32_t height, uint32_t k0, uint32_t k1, uint64_t n,
                                                       const uint32_t* px_i, const uint32_t* px_j, const uint32_t* sample, rt_ray* out) {
    uint64_t k = (uint64_t)blockIdx.x * RT_BS + threadIdx.x;
    if (k >= n) return;
    Ray r = camera_ray(cam, px_i[k], px_j[k], width, height, sample[k], k0, k1);
    rt_ray o;
    o.origin[0] = r.o.x; o.origin[1] = r.o.y; o.origin[2] = r.o.z;
    o.direction[0] = r.d.x; o.direction[1] = r.d.y; o.direction[2] = r.d.z;
    o.time = r.time;
    out[k] = o;
}

// -------------------------------------------------------------------- host
static thread_local std::string g_err;
static rt_status fail(rt_status s, const std::string& m) { g_err = m; return s; }
#define HIP_TRY(expr)                                                                                          \
    do {                                                                                                       \
        hipError_t e_ = (expr);                                                                                \
        if (e_ != hipSuccess) {                                                                                \
            return fail(e_ == hipErrorOutOfMemory ? RT_ERR_OUT_OF_MEMORY : RT_ERR_HIP,                         \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                                    \
        }                                                                                                      \
    } while (0)

#ifndef RT_AUTO_PASS_PATHS
#define RT_AUTO_PASS_PATHS (128ull << 20)  // paths of a pass when the caller leaves samples_per_pass at 0 (120 bytes of workspace each)
#endif
#ifndef RT_PASS_PARTS
#define RT_PASS_PARTS 2   // parts a pass is split into when splitting pays (each on a stream of its own)
#endif
#define RT_MAX_PARTS 4
// Device memory of a render beyond its inputs and outputs: sized for one pass of `paths` paths over `pixels` owned pixels.
struct Workspace {
    uint64_t paths = 0, pixels = 0;
    uint32_t n_counts = 0;          // queue lengths: RT_MAX_PARTS x 2 arrays of this many entries
    RayQueue q[2]{};
    uint2* hits = nullptr;
    float4* values = nullptr;
    double* accum = nullptr;
    uint32_t* owned = nullptr;
    uint32_t* counts = nullptr;
};
// What the first pass of a frame learns by reading queue lengths back and timing its kernels, kept per frame shape so
// that repeat frames (and every frame of a rank with a small share) run without a host round trip.  It only schedules:
// the image does not depend on it.  The bounce at which the queue is short enough for the tail kernel is learnt against an
// absolute length, so it belongs to the pass size it was seen at (`pass_paths`: a frame of one sample per pixel hands over
// after a bounce, the same frame in passes of 64 M paths after ten), and so does the comparison of the two kernels' times.
struct Calibration { uint32_t width, height, depth, n_owned; uint64_t pass_paths; int tail_from; bool split_pays; };
struct rt_scene {
    std::mutex mutex;  // held by every entry point for the duration of a call: rendering changes the handle (workspace, schedule, tree, profile)
    int device = 0;
    uint32_t n_cu = 0;
    uint32_t n_leaves = 0;  // leaves of the tree (primitives that can be hit)
    uint32_t n_long_leaves = 0;  // ... of which tested by the general primitive test (wrapped shapes, media)
    SceneView view{};
    std::vector<void*> allocs;
    hipStream_t own_stream = nullptr, side_stream[RT_MAX_PARTS - 1] = {};  // side streams: the other parts of a pass (see render_impl)
    hipEvent_t ev_fork = nullptr, ev_join[RT_MAX_PARTS - 1] = {};
    hipEvent_t ev_in = nullptr, ev_out = nullptr;  // order a caller's stream with the library's own (render_impl)
    bool nodes_fit_lds = false;
    Workspace ws;
    uint32_t* h_counts = nullptr;  // pinned
    std::vector<void*> ws_allocs;
    std::vector<Calibration> calibrations;
    // what make_walk_nodes reads of the caller's description, kept for regrouping the tree by sampled rays (tune_tree)
    struct HostTree { std::vector<rt_prim> prims; std::vector<rt_sphere> spheres; std::vector<rt_rect> rects; std::vector<rt_box> boxes;
                      std::vector<rt_xform> xforms; std::vector<rt_bvh_node> bvh; rt_scene_desc d{}; } host;
    bool tree_tuned = false;
    double tree_tests_by_area = 0.0, tree_tests_sampled = 0.0;
    float* d_tune_rays = nullptr;  // RT_TUNE_BOUNCES x RT_TUNE_PER_BOUNCE x 6 floats
    // profile
    rt_profile prof{};
    std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> events;
    std::vector<hipEvent_t> event_pool;
};

// ---- the workspace of rt_render_device_noalloc: one array per GPU, in the library's image, never allocated or freed.
// Largest pass and frame share it can hold (the entry point rejects more):
#define RT_STATIC_MAX_PATHS (1u << 18)   // paths per pass
#define RT_STATIC_MAX_PIXELS (1u << 18)  // pixels this rank owns
#define RT_STATIC_MAX_DEPTH 254u         // max_reflections (queue-length slots: depth + 2)
constexpr size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
constexpr size_t ST_QUEUE = align256((size_t)RT_STATIC_MAX_PATHS * sizeof(float4));
constexpr size_t ST_OFF_HITS = 6 * ST_QUEUE;
constexpr size_t ST_OFF_VALUES = ST_OFF_HITS + align256((size_t)RT_STATIC_MAX_PATHS * sizeof(uint2));
constexpr size_t ST_OFF_ACCUM = ST_OFF_VALUES + ST_QUEUE;
constexpr size_t ST_OFF_OWNED = ST_OFF_ACCUM + align256((size_t)RT_STATIC_MAX_PIXELS * 3 * sizeof(double));
constexpr size_t ST_OFF_COUNTS = ST_OFF_OWNED + align256((size_t)RT_STATIC_MAX_PIXELS * sizeof(uint32_t));
constexpr size_t ST_BYTES = ST_OFF_COUNTS + align256((size_t)2 * (RT_STATIC_MAX_DEPTH + 2) * sizeof(uint32_t));  // about 38.5 MB
static __device__ __align__(256) unsigned char scratch[ST_BYTES];
// `scratch` above is a host-side stand-in for the symbol, not an address: each GPU has its own copy at its own address,
// looked up once per device (rt_scene_create, with that device current) and used from then on.
#define RT_MAX_DEVICES 64
static unsigned char* g_scratch_ptr[RT_MAX_DEVICES] = {};
static std::mutex g_scratch_mutex;

template <class T>
static rt_status upload(rt_scene* sc, const T* src, size_t n, const T** dst) {
    *dst = nullptr;
    if (n == 0) return RT_OK;
    void* d = nullptr;
    HIP_TRY(hipMalloc(&d, n * sizeof(T)));
    sc->allocs.push_back(d);
    HIP_TRY(hipMemcpy(d, src, n * sizeof(T), hipMemcpyHostToDevice));
    *dst = (const T*)d;
    return RT_OK;
}

static rt_status validate_desc(const rt_scene_desc* D) {
    if (!D) return fail(RT_ERR_INVALID_ARGUMENT, "scene description is null");
    if (D->abi_version != RT_ABI_VERSION) return fail(RT_ERR_INVALID_ARGUMENT, "abi_version mismatch");
    if (D->n_prims == 0) return fail(RT_ERR_EMPTY_SCENE, "a scene needs at least one object (Scene::new asserts this)");
    if (D->n_prims >= (1u << 21)) return fail(RT_ERR_UNSUPPORTED, "too many primitives");
    if (!D->prims || !D->materials || D->n_materials == 0) return fail(RT_ERR_INVALID_ARGUMENT, "prims / materials missing");
    if ((D->n_spheres && !D->spheres) || (D->n_rects && !D->rects) || (D->n_boxes && !D->boxes) || (D->n_xforms && !D->xforms) ||
        (D->n_textures && !D->textures) || (D->n_images && !D->images) || (D->n_perm_tables && !D->perm_tables) || (D->n_bvh_nodes && !D->bvh))
        return fail(RT_ERR_INVALID_ARGUMENT, "an array with a non-zero count is null");
    for (uint32_t i = 0; i < D->n_textures; ++i) {
        const rt_texture& T = D->textures[i];
        if (T.kind > RT_TEX_IMAGE) return fail(RT_ERR_INVALID_ARGUMENT, "texture kind out of range");
        if (T.kind == RT_TEX_CHECKERED && (T.odd >= D->n_textures || T.even >= D->n_textures)) return fail(RT_ERR_INVALID_ARGUMENT, "checker child out of range");
        if (T.kind == RT_TEX_NOISE) {
            const uint32_t fn = T.odd & 255u, octaves = (T.odd >> 8) & 255u;
            if (fn > RT_NOISE_TURBULENCE) return fail(RT_ERR_UNSUPPORTED, "noise texture names a function the kernels do not implement");
            if (fn != RT_NOISE_PERLIN && (octaves < 1 || octaves > 32)) return fail(RT_ERR_INVALID_ARGUMENT, "a noise texture's octave count must be 1..32");
            const uint64_t tables = fn == RT_NOISE_PERLIN ? 1u : (fn == RT_NOISE_FBM ? octaves : octaves + 2u);  // consecutive tables from `resource` on
            if ((uint64_t)T.resource + tables > D->n_perm_tables || (fn == RT_NOISE_TURBULENCE && T.even >= D->n_perm_tables))
                return fail(RT_ERR_INVALID_ARGUMENT, "noise permutation table out of range");
        }
        if (T.kind == RT_TEX_IMAGE) {
            if (T.resource >= D->n_images) return fail(RT_ERR_INVALID_ARGUMENT, "image index out of range");
            const rt_image& I = D->images[T.resource];
            if (I.width == 0 || I.height == 0 || !D->texels || I.texel_offset + 3ull * I.width * I.height > D->texel_bytes)
                return fail(RT_ERR_INVALID_ARGUMENT, "image texels out of range");
        }
    }
    // checker nesting: children must resolve to a leaf texture within RT_MAX_TEXTURE_NESTING levels on every branch
    for (uint32_t i = 0; i < D->n_textures; ++i) {
        std::vector<std::pair<uint32_t, uint32_t>> todo{{i, 0}};
        size_t visited = 0;
        while (!todo.empty()) {
            auto [t, depth] = todo.back(); todo.pop_back();
            if (++visited > 4096) return fail(RT_ERR_UNSUPPORTED, "checker texture tree too large");
            if (D->textures[t].kind != RT_TEX_CHECKERED) continue;
            if (depth >= RT_MAX_TEXTURE_NESTING) return fail(RT_ERR_UNSUPPORTED, "checker textures nested too deeply (or cyclic)");
            todo.push_back({D->textures[t].odd, depth + 1}); todo.push_back({D->textures[t].even, depth + 1});
        }
    }
    for (uint32_t i = 0; i < D->n_materials; ++i) {
        const rt_material& M = D->materials[i];
        if (M.kind > RT_MAT_ISOTROPIC) return fail(RT_ERR_INVALID_ARGUMENT, "material kind out of range");
        bool textured = M.kind == RT_MAT_LAMBERTIAN || M.kind == RT_MAT_DIFFUSE_LIGHT || M.kind == RT_MAT_ISOTROPIC;
        if (textured && M.texture >= D->n_textures) return fail(RT_ERR_INVALID_ARGUMENT, "material texture out of range");
    }
    for (uint32_t i = 0; i < D->n_prims; ++i) {
        const rt_prim& P = D->prims[i];
        uint32_t lim = P.geom <= RT_GEOM_MOVING_SPHERE ? D->n_spheres : P.geom == RT_GEOM_RECT ? D->n_rects : P.geom == RT_GEOM_BOX ? D->n_boxes : 0;
        if (P.geom > RT_GEOM_BOX || P.geom_index >= lim) return fail(RT_ERR_INVALID_ARGUMENT, "primitive geometry out of range");
        if (P.material >= D->n_materials) return fail(RT_ERR_INVALID_ARGUMENT, "primitive material out of range");
        if (P.xf_count > RT_MAX_XFORMS_PER_PRIM) return fail(RT_ERR_UNSUPPORTED, "too many nested Translated/Rotated wrappers on one shape");
        if ((uint64_t)P.xf_first + P.xf_count > D->n_xforms) return fail(RT_ERR_INVALID_ARGUMENT, "primitive transform range out of bounds");
        if (P.xf_outer > P.xf_count) return fail(RT_ERR_INVALID_ARGUMENT, "xf_outer exceeds xf_count");
        if (P.geom == RT_GEOM_RECT && D->rects[P.geom_index].axis > 2) return fail(RT_ERR_INVALID_ARGUMENT, "rectangle axis out of range");
    }
    for (uint32_t i = 0; i < D->n_xforms; ++i)
        if (D->xforms[i].kind > RT_XF_ROTATE_Z) return fail(RT_ERR_INVALID_ARGUMENT, "transform kind out of range");
    // BVH: pre-order with skip links.  skip > index makes every walk terminate; the nesting check makes the
    // array a tree; every non-inert primitive sits in exactly one leaf.
    std::vector<uint8_t> seen_prim(D->n_prims, 0);
    {
        std::vector<uint32_t> open_ends;  // skip values of the inner nodes we are inside of
        for (uint32_t i = 0; i < D->n_bvh_nodes; ++i) {
            const rt_bvh_node& N = D->bvh[i];
            while (!open_ends.empty() && open_ends.back() == i) open_ends.pop_back();
            if (N.skip <= i || N.skip > D->n_bvh_nodes) return fail(RT_ERR_INVALID_ARGUMENT, "BVH skip link does not move forward");
            if (!open_ends.empty() && N.skip > open_ends.back()) return fail(RT_ERR_INVALID_ARGUMENT, "BVH subtree crosses its parent's end");
            if (N.prim != RT_BVH_NONE) {
                if (N.skip != i + 1) return fail(RT_ERR_INVALID_ARGUMENT, "BVH leaf with a subtree");
                const uint32_t pr = N.prim & ~RT_BVH_EXACT;
                if (pr >= D->n_prims || seen_prim[pr]) return fail(RT_ERR_INVALID_ARGUMENT, "BVH leaf primitive out of range or repeated");
                seen_prim[pr] = 1;
            } else {
                if (N.skip == i + 1) return fail(RT_ERR_INVALID_ARGUMENT, "BVH inner node without children");
                open_ends.push_back(N.skip);
            }
        }
    }
    for (uint32_t i = 0; i < D->n_prims; ++i)
        if (!seen_prim[i] && !(D->prims[i].flags & RT_PRIM_INERT)) return fail(RT_ERR_INVALID_ARGUMENT, "a primitive is missing from the BVH");
    return RT_OK;
}

static void free_workspace(rt_scene* sc) {
    for (void* p : sc->ws_allocs) (void)hipFree(p);
    sc->ws_allocs.clear();
    if (sc->h_counts) { (void)hipHostFree(sc->h_counts); sc->h_counts = nullptr; }
    sc->ws = Workspace{};
}

// This device's address of `scratch` (the device is current).
static rt_status resolve_scratch(int device) {
    if (device < 0 || device >= RT_MAX_DEVICES) return RT_OK;  // (rt_render_device_noalloc then reports that the device has no array)
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch_ptr[device]) return RT_OK;
    void* p = nullptr;
    HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    g_scratch_ptr[device] = static_cast<unsigned char*>(p);
    return RT_OK;
}

extern "C" {

uint32_t rt_abi_version(void) { return RT_ABI_VERSION; }
uint64_t rt_auto_pass_paths(void) { return (uint64_t)RT_AUTO_PASS_PATHS; }
const char* rt_last_error(void) { return g_err.c_str(); }
int rt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void rt_scene_destroy(rt_scene* sc) {
    if (!sc) return;
    (void)hipSetDevice(sc->device);
    free_workspace(sc);
    for (void* p : sc->allocs) (void)hipFree(p);
    for (auto& e : sc->event_pool) (void)hipEventDestroy(e);
    if (sc->own_stream) (void)hipStreamDestroy(sc->own_stream);
    for (auto& x : sc->side_stream) if (x) (void)hipStreamDestroy(x);
    if (sc->ev_fork) (void)hipEventDestroy(sc->ev_fork);
    if (sc->ev_in) (void)hipEventDestroy(sc->ev_in);
    if (sc->ev_out) (void)hipEventDestroy(sc->ev_out);
    for (auto& x : sc->ev_join) if (x) (void)hipEventDestroy(x);
    delete sc;
}

rt_status rt_scene_create(const rt_scene_desc* D, int device, rt_scene** out) {
    if (!out) return fail(RT_ERR_INVALID_ARGUMENT, "out is null");
    *out = nullptr;
    rt_status st = validate_desc(D);
    if (st != RT_OK) return st;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(RT_ERR_NO_DEVICE, "no HIP device is available; this library has no CPU path");
    if (device < 0) { if (hipGetDevice(&device) != hipSuccess) device = 0; }
    if (device >= ndev) return fail(RT_ERR_NO_DEVICE, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    rt_scene* sc = new rt_scene();
    sc->device = device;
    SceneView& V = sc->view;
#define UP(field, src, n) do { st = upload(sc, src, (size_t)(n), &V.field); if (st != RT_OK) { rt_scene_destroy(sc); return st; } } while (0)
    UP(prims, D->prims, D->n_prims);
    UP(spheres, D->spheres, D->n_spheres);
    {   // rectangles and boxes in the kernels' own form (fp32 value + fp32 remainder per f64 coordinate)
        std::vector<DevRect> rects; std::vector<DevBox> boxes;
        make_device_shapes(D, rects, boxes);
        UP(rects, rects.data(), rects.size());
        UP(boxes, boxes.data(), boxes.size());
    }
    UP(xforms, D->xforms, D->n_xforms);
    {   // the primitives again, each with its material beside it (flags set, solid colours inlined)
        std::vector<ShadeRec> recs;
        make_shade_records(D, recs);
        UP(shade, recs.data(), recs.size());
    }
    UP(textures, D->textures, D->n_textures);
    UP(images, D->images, D->n_images);
    UP(texels, D->texels, D->texel_bytes);
    UP(perm, D->perm_tables, 256u * (size_t)D->n_perm_tables);
    {   // what a later regrouping of the tree needs of the description (the caller's arrays are only valid during this call)
        rt_scene::HostTree& H = sc->host;
        H.prims.assign(D->prims, D->prims + D->n_prims);
        if (D->n_spheres) H.spheres.assign(D->spheres, D->spheres + D->n_spheres);
        if (D->n_rects) H.rects.assign(D->rects, D->rects + D->n_rects);
        if (D->n_boxes) H.boxes.assign(D->boxes, D->boxes + D->n_boxes);
        if (D->n_xforms) H.xforms.assign(D->xforms, D->xforms + D->n_xforms);
        if (D->n_bvh_nodes) H.bvh.assign(D->bvh, D->bvh + D->n_bvh_nodes);
        H.d = rt_scene_desc{};
        H.d.n_prims = D->n_prims; H.d.n_spheres = D->n_spheres; H.d.n_rects = D->n_rects; H.d.n_boxes = D->n_boxes; H.d.n_xforms = D->n_xforms;
        H.d.n_bvh_nodes = D->n_bvh_nodes;
        H.d.prims = H.prims.data(); H.d.spheres = H.spheres.data(); H.d.rects = H.rects.data(); H.d.boxes = H.boxes.data(); H.d.xforms = H.xforms.data();
        H.d.bvh = H.bvh.data();
        void* p = nullptr;
        if (hipMalloc(&p, (size_t)RT_TUNE_BOUNCES * RT_TUNE_PER_BOUNCE * 6 * sizeof(float)) != hipSuccess) { rt_scene_destroy(sc); return fail(RT_ERR_OUT_OF_MEMORY, "hipMalloc failed"); }
        sc->allocs.push_back(p); sc->d_tune_rays = static_cast<float*>(p);
    }
    {   // the walk's own node array (payload slots, leaf annotations, sentinel)
        std::vector<WalkNode> nodes;
        make_walk_nodes(D, nodes);
        for (size_t pos = 0; pos + 1 < nodes.size();) {  // leaves whose test is the general one (see scan_queue_end); a leaf's skip link jumps over its payload slots
            const WalkNode& N = nodes[pos];
            if (N.prim < RT_NODE_LEAF) { ++pos; continue; }
            if (!(N.prim & (RT_NODE_PLAIN_SPHERE | RT_NODE_PLAIN_BOX | RT_NODE_PLAIN_RECT))) ++sc->n_long_leaves;
            pos = std::max<size_t>(N.skip / RT_NODE_STRIDE, pos + 1);
        }
        if (nodes.size() * sizeof(WalkNode) >= RT_NODE_LEAF) {  // node positions are byte offsets and share a word with the leaf bits
            rt_scene_destroy(sc);
            return fail(RT_ERR_UNSUPPORTED, "BVH too large: node positions must stay below 2^28 bytes");
        }
        UP(bvh, nodes.data(), nodes.size());
        V.n_bvh_nodes = (uint32_t)nodes.size() - 1;
    }
#undef UP
    for (uint32_t i = 0; i < D->n_bvh_nodes; ++i) sc->n_leaves += D->bvh[i].prim != RT_BVH_NONE;
    V.n_prims = D->n_prims;
    V.n_spheres = D->n_spheres; V.n_rects = D->n_rects; V.n_boxes = D->n_boxes; V.n_xforms = D->n_xforms;
    V.n_textures = D->n_textures; V.n_images = D->n_images; V.n_perm_tables = D->n_perm_tables;
    V.lds_scene_bytes = 0;
    V.side_rounding = scene_needs_side_rounding(D) ? 1u : 0u;
    V.fp32_planes = scene_tree_holds_a_medium(D) ? 0u : 1u;
    V.multi_scatter = scene_has_multi_scatter_media(D) ? 1u : 0u;
    for (int c = 0; c < 3; ++c) V.background[c] = (float)D->background[c];
    hipError_t e = hipStreamCreateWithFlags(&sc->own_stream, hipStreamNonBlocking);
    for (auto& x : sc->side_stream) if (e == hipSuccess) e = hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&sc->ev_fork, hipEventDisableTiming);
    for (auto& x : sc->ev_join) if (e == hipSuccess) e = hipEventCreateWithFlags(&x, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&sc->ev_in, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&sc->ev_out, hipEventDisableTiming);
    if (e != hipSuccess) { rt_scene_destroy(sc); return fail(RT_ERR_HIP, std::string("stream / event creation: ") + hipGetErrorString(e)); }
    // The node array goes to LDS when it fits beside the workgroup's ray counter (160 KB per CU, one 1024-thread workgroup
    // per CU): the walk kernels then take the start of their dynamic LDS for address 0 (walk_step_flat).
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) { rt_scene_destroy(sc); return fail(RT_ERR_HIP, std::string("hipGetDeviceProperties: ") + hipGetErrorString(e)); }
    sc->n_cu = (uint32_t)prop.multiProcessorCount;
    const size_t node_bytes = ((size_t)V.n_bvh_nodes + 1) * sizeof(WalkNode);
    sc->nodes_fit_lds = V.n_bvh_nodes > 0 && node_bytes + RT_LDS_BEHIND_NODES <= RT_LDS_NODE_LIMIT;
    {   // a small scene's tables go to LDS with the nodes (stage_nodes): sized exactly as stage_table lays them out
        auto a16 = [](size_t b) { return (b + 15) & ~(size_t)15; };
        const size_t tables = a16(sizeof(rt_prim) * D->n_prims) + a16(sizeof(rt_sphere) * D->n_spheres) + a16(sizeof(DevRect) * D->n_rects) +
                              a16(sizeof(DevBox) * D->n_boxes) + a16(sizeof(rt_xform) * D->n_xforms) + a16(sizeof(ShadeRec) * D->n_prims) +
                              a16(sizeof(rt_texture) * D->n_textures) + a16(sizeof(rt_image) * D->n_images) + a16(256u * (size_t)D->n_perm_tables);
        if (sc->nodes_fit_lds && tables <= RT_LDS_SCENE_LIMIT && node_bytes + RT_LDS_BEHIND_NODES + tables <= RT_LDS_NODE_LIMIT) V.lds_scene_bytes = (uint32_t)tables;
    }
    if (sc->nodes_fit_lds) {
        hipFuncAttributes fa;
        for (const void* f : {reinterpret_cast<const void*>(&k_traverse<true, false>), reinterpret_cast<const void*>(&k_traverse<true, true>),
                              reinterpret_cast<const void*>(&k_traverse<true, false, false>)}) {
            if (e == hipSuccess) e = hipFuncGetAttributes(&fa, f);
            if (e == hipSuccess && fa.sharedSizeBytes != 0) { rt_scene_destroy(sc); return fail(RT_ERR_HIP, "k_traverse has static LDS in front of its node array"); }
            if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, RT_LDS_NODE_LIMIT);
        }
        const void* const walkers[] = {reinterpret_cast<const void*>(&k_tail<true, false>), reinterpret_cast<const void*>(&k_tail<true, true>),
                                       reinterpret_cast<const void*>(&k_tail<true, false, false>), reinterpret_cast<const void*>(&k_tail<true, true, false>),
                                       reinterpret_cast<const void*>(&k_shade<SHADE_WALK, false>), reinterpret_cast<const void*>(&k_shade<SHADE_WALK, true>),
                                       reinterpret_cast<const void*>(&k_shade<SHADE_SCAN, false>), reinterpret_cast<const void*>(&k_shade<SHADE_SCAN, true>),
                                       reinterpret_cast<const void*>(&k_shade<SHADE_SCAN, false, true>), reinterpret_cast<const void*>(&k_shade<SHADE_SCAN, true, true>),
                                       reinterpret_cast<const void*>(&k_shade<SHADE_SCAN, false, false, true>), reinterpret_cast<const void*>(&k_shade<SHADE_SCAN, true, false, true>),
                                       reinterpret_cast<const void*>(&k_shade<SHADE_SCAN, false, true, true>), reinterpret_cast<const void*>(&k_shade<SHADE_SCAN, true, true, true>),
                                       reinterpret_cast<const void*>(&k_shade<SHADE_SCAN, false, false, true, true>), reinterpret_cast<const void*>(&k_shade<SHADE_SCAN, true, false, true, true>),
                                       reinterpret_cast<const void*>(&k_shade<SHADE_SCAN, false, true, false, false, 1>), reinterpret_cast<const void*>(&k_shade<SHADE_SCAN, true, true, false, false, 1>),
                                       reinterpret_cast<const void*>(&k_shade<SHADE_SCAN, false, true, false, false, 2>), reinterpret_cast<const void*>(&k_shade<SHADE_SCAN, true, true, false, false, 2>),
                                       };
        for (const void* f : walkers) if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, RT_LDS_NODE_LIMIT);
        if (e != hipSuccess) { rt_scene_destroy(sc); return fail(RT_ERR_HIP, std::string("LDS set-up of the walk kernels: ") + hipGetErrorString(e)); }
    }
    if ((st = resolve_scratch(device)) != RT_OK) { rt_scene_destroy(sc); return st; }
    *out = sc;
    return RT_OK;
}

}  // extern "C"

static rt_status ensure_workspace(rt_scene* sc, uint64_t n_paths, uint64_t n_pixels, uint32_t n_counts) {
    Workspace& w = sc->ws;
    if (n_paths <= w.paths && n_pixels <= w.pixels && n_counts <= w.n_counts) return RT_OK;
    free_workspace(sc);
    n_paths = std::max<uint64_t>(n_paths, 1); n_pixels = std::max<uint64_t>(n_pixels, 1);
    auto alloc = [&](void** p, size_t bytes) -> rt_status {
        HIP_TRY(hipMalloc(p, bytes));
        sc->ws_allocs.push_back(*p);
        return RT_OK;
    };
    rt_status st;
    for (int k = 0; k < 2; ++k) {
        if ((st = alloc((void**)&w.q[k].od0, n_paths * sizeof(float4))) != RT_OK) return st;
        if ((st = alloc((void**)&w.q[k].od1, n_paths * sizeof(float4))) != RT_OK) return st;
        if ((st = alloc((void**)&w.q[k].thr, n_paths * sizeof(float4))) != RT_OK) return st;
    }
    if ((st = alloc((void**)&w.hits, n_paths * sizeof(uint2))) != RT_OK) return st;
    if ((st = alloc((void**)&w.values, n_paths * sizeof(float4))) != RT_OK) return st;
    if ((st = alloc((void**)&w.accum, n_pixels * 3 * sizeof(double))) != RT_OK) return st;
    if ((st = alloc((void**)&w.owned, n_pixels * sizeof(uint32_t))) != RT_OK) return st;
    if ((st = alloc((void**)&w.counts, RT_MAX_PARTS * 2 * (size_t)n_counts * sizeof(uint32_t))) != RT_OK) return st;  // per part of a pass: front and back lengths
    HIP_TRY(hipHostMalloc((void**)&sc->h_counts, 4 * sizeof(uint32_t), hipHostMallocDefault));
    w.paths = n_paths; w.pixels = n_pixels; w.n_counts = n_counts;
    return RT_OK;
}

// The same layout inside this device's `scratch` array.
static Workspace static_workspace(unsigned char* base) {
    Workspace w;
    w.paths = RT_STATIC_MAX_PATHS; w.pixels = RT_STATIC_MAX_PIXELS; w.n_counts = RT_STATIC_MAX_DEPTH + 2;
    for (int k = 0; k < 2; ++k) {
        w.q[k].od0 = reinterpret_cast<float4*>(base + (3 * k + 0) * ST_QUEUE);
        w.q[k].od1 = reinterpret_cast<float4*>(base + (3 * k + 1) * ST_QUEUE);
        w.q[k].thr = reinterpret_cast<float4*>(base + (3 * k + 2) * ST_QUEUE);
    }
    w.hits = reinterpret_cast<uint2*>(base + ST_OFF_HITS);
    w.values = reinterpret_cast<float4*>(base + ST_OFF_VALUES);
    w.accum = reinterpret_cast<double*>(base + ST_OFF_ACCUM);
    w.owned = reinterpret_cast<uint32_t*>(base + ST_OFF_OWNED);
    w.counts = reinterpret_cast<uint32_t*>(base + ST_OFF_COUNTS);
    return w;
}

namespace {
struct Launcher {
    rt_scene* sc; hipStream_t stream; bool profile;
    rt_status begin(int kind, hipEvent_t* stop_out) {
        *stop_out = nullptr;
        if (!profile) return RT_OK;
        hipEvent_t ev[2];
        for (int k = 0; k < 2; ++k) {
            if (!sc->event_pool.empty()) { ev[k] = sc->event_pool.back(); sc->event_pool.pop_back(); }
            else HIP_TRY(hipEventCreate(&ev[k]));
        }
        HIP_TRY(hipEventRecord(ev[0], stream));
        sc->events.push_back({kind, {ev[0], ev[1]}});
        *stop_out = ev[1];
        return RT_OK;
    }
    rt_status end(hipEvent_t stop) {
        HIP_TRY(hipGetLastError());
        if (stop) HIP_TRY(hipEventRecord(stop, stream));
        return RT_OK;
    }
};
}  // namespace

static rt_status collect_profile(rt_scene* sc) {
    for (auto& e : sc->events) {
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, e.second.first, e.second.second));
        sc->prof.kernel_ms[e.first] += ms;
        sc->prof.kernel_launches[e.first] += 1;
        sc->event_pool.push_back(e.second.first); sc->event_pool.push_back(e.second.second);
    }
    sc->events.clear();
    return RT_OK;
}

static rt_status validate_render(rt_scene* sc, const rt_camera* cam, const rt_image_options* io, const rt_render_options* ro) {
    if (!sc || !cam || !io || !ro) return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
    if (io->width < 2 || io->height < 2) return fail(RT_ERR_INVALID_ARGUMENT, "width and height must be at least 2 (pixel() divides by width-1 and height-1)");
    if ((uint64_t)io->width * io->height >= (1ull << 31)) return fail(RT_ERR_UNSUPPORTED, "image too large");
    if (io->samples_per_pixel == 0 || io->samples_per_pixel >= (1ull << 32)) return fail(RT_ERR_INVALID_ARGUMENT, "samples_per_pixel must be in [1, 2^32)");
    if (io->max_reflections >= (1ull << 31)) return fail(RT_ERR_INVALID_ARGUMENT, "max_reflections too large");
    if (!(cam->shutter_open < cam->shutter_close)) return fail(RT_ERR_INVALID_ARGUMENT, "shutter_open must be below shutter_close (gen_range panics otherwise)");
    if (ro->world_size == 0 || ro->rank >= ro->world_size) return fail(RT_ERR_INVALID_ARGUMENT, "rank / world_size invalid");
    return RT_OK;
}

// Renders this rank's pixels into d_mean_out (full frame of doubles on the device).
// `fixed`: the workspace of the variant that allocates nothing (null: the scene handle's own, grown as needed).  That
// variant also creates no events and reads nothing back: it hands over to k_tail at a bounce fixed in advance.
// Box tests a ray makes on its way through a node array when nothing it meets stops it (no primitive tests: every leaf it
// reaches counts as passed): what regrouping the inner nodes changes, for the statistics of rt_profile.
static double node_tests_per_ray(const std::vector<WalkNode>& nodes, const SampleRays& rays) {
    if (nodes.size() < 2 || rays.size() == 0) return 0.0;
    const uint32_t end = (uint32_t)(nodes.size() - 1) * RT_NODE_STRIDE;
    uint64_t tests = 0;
    for (size_t k = 0; k < rays.size(); ++k) {
        const V3 inv = walk_cap(v3(rays.inv[3 * k], rays.inv[3 * k + 1], rays.inv[3 * k + 2]));
        const V3 oi = v3(rays.o[3 * k] * inv.x, rays.o[3 * k + 1] * inv.y, rays.o[3 * k + 2] * inv.z);
        for (uint32_t i = 0; i < end;) {
            const WalkNode& N = node_at(nodes.data(), i);
            ++tests;
            i = (node_hit(N, inv, oi, 0.001f, RT_WALK_FAR) && N.prim < RT_NODE_LEAF) ? N.prim : N.skip;
        }
    }
    return (double)tests / (double)rays.size();
}

// Regroups the inner nodes of the library's copy of the tree by a sample of the rays a pass has traced (d_tune_rays, filled
// by k_sample_rays; the stream is idle).  Leaves, their order and their boxes stay, so no hit and no image changes; the walk
// meets fewer boxes (final_scene: 27 instead of 34 a segment).  Done once per scene handle.  Failure to improve, or a node
// array that no longer fits where the old one did, leaves everything as it was.
static rt_status tune_tree(rt_scene* sc, uint32_t n_bounces) {
    std::vector<float> h((size_t)n_bounces * RT_TUNE_PER_BOUNCE * 6);
    HIP_TRY(hipMemcpy(h.data(), sc->d_tune_rays, h.size() * sizeof(float), hipMemcpyDeviceToHost));
    SampleRays rays;
    for (size_t k = 0; k < h.size() / 6; ++k) {
        const float* r = &h[6 * k];
        bool ok = true;
        for (int c = 0; c < 6; ++c) ok = ok && r[c] - r[c] == 0.0f;
        if (ok && (r[3] != 0.0f || r[4] != 0.0f || r[5] != 0.0f)) rays.add(r, r + 3);
    }
    if (rays.size() < 64) return RT_OK;  // (too few: a later, larger pass may lend more)
    sc->tree_tuned = true;
    std::vector<WalkNode> before, after;
    make_walk_nodes(&sc->host.d, before);
    make_walk_nodes(&sc->host.d, after, &rays);
    const double t_before = node_tests_per_ray(before, rays), t_after = node_tests_per_ray(after, rays);
    sc->tree_tests_by_area = t_before; sc->tree_tests_sampled = t_before;
    const size_t bytes = after.size() * sizeof(WalkNode);
    const bool fits = !sc->nodes_fit_lds || bytes + RT_LDS_BEHIND_NODES + sc->view.lds_scene_bytes <= RT_LDS_NODE_LIMIT;
    if (!(t_after < t_before) || !fits || bytes >= RT_NODE_LEAF) return RT_OK;
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); return RT_OK; }
    sc->allocs.push_back(p);
    HIP_TRY(hipMemcpy(p, after.data(), bytes, hipMemcpyHostToDevice));
    sc->view.bvh = static_cast<const WalkNode*>(p);  // (the old array stays allocated until the handle goes: nothing reads it any more)
    sc->view.n_bvh_nodes = (uint32_t)after.size() - 1;
    sc->tree_tests_sampled = t_after;
    return RT_OK;
}

static rt_status render_impl(rt_scene* sc, const rt_camera* cam, const rt_image_options* io, const rt_render_options* ro, double* d_mean_out, bool sync_at_end,
                             const Workspace* fixed) {
    rt_status st = validate_render(sc, cam, io, ro);
    if (st != RT_OK) return st;
    if (!d_mean_out) return fail(RT_ERR_INVALID_ARGUMENT, "output buffer is null");
    HIP_TRY(hipSetDevice(sc->device));
    // The work always runs on the library's own streams; a caller's stream only orders it: the render starts after what the
    // caller has queued there and the caller's stream continues after the frame is complete.  (The two halves of a pass
    // overlap only if their streams sit on different hardware queues, which the library's two streams, created one after
    // the other, do; run on a stream made elsewhere, final_scene lost 13 % to serialisation.)
    hipStream_t caller = ro->stream ? (hipStream_t)ro->stream : nullptr;
    hipStream_t stream = sc->own_stream;
    if (caller) {
        // A stream of raised priority must not wait for work that streams of normal priority have yet to do: on this runtime
        // its pending wait keeps them from being scheduled and neither side ever finishes (seen with a priority -1 stream of
        // torch's pool behind a frame still being rendered: hipStreamSynchronize on it never returned; the same chain through a
        // stream of priority 0 completes).  The caller's stream waits for the library's, which have normal priority.
        int priority = 0;
        HIP_TRY(hipStreamGetPriority(caller, &priority));
        if (priority < 0) return fail(RT_ERR_UNSUPPORTED, "rt_render_options.stream has a raised priority: it would wait for the library's normal-priority streams, which this runtime never schedules past such a wait; pass a stream of default priority");
        HIP_TRY(hipEventRecord(sc->ev_in, caller));
        HIP_TRY(hipStreamWaitEvent(stream, sc->ev_in, 0));
    }
    auto hand_back = [&]() -> rt_status {
        if (caller) {
            HIP_TRY(hipEventRecord(sc->ev_out, stream));
            HIP_TRY(hipStreamWaitEvent(caller, sc->ev_out, 0));
        }
        return RT_OK;
    };
    const bool profile = (ro->flags & RT_RENDER_PROFILE) != 0;
    if (fixed && profile) return fail(RT_ERR_INVALID_ARGUMENT, "RT_RENDER_PROFILE creates events: not available without allocation");
    const uint32_t W = io->width, H = io->height;
    // (a tile larger than the frame is the frame)
    const TileGrid tiles{W, H, std::min(ro->tile_w ? ro->tile_w : 16u, W), std::min(ro->tile_h ? ro->tile_h : 16u, H), ro->rank, ro->world_size};
    if (tiles.ntiles() >= (1ull << 31)) return fail(RT_ERR_UNSUPPORTED, "too many tiles");
    const uint64_t spp = io->samples_per_pixel;
    const uint32_t depth = (uint32_t)io->max_reflections;
    const uint32_t n_counts = depth + 2;
    const uint64_t n_owned64 = owned_pixels_below(tiles, tiles.ntiles());
    if (fixed && (n_owned64 > fixed->pixels || n_counts > fixed->n_counts)) {
        char msg[200];
        snprintf(msg, sizeof(msg), "the fixed workspace holds at most %llu pixels per rank and max_reflections %u (this frame: %llu pixels, %u)",
                 (unsigned long long)fixed->pixels, fixed->n_counts - 2u, (unsigned long long)n_owned64, depth);
        return fail(RT_ERR_UNSUPPORTED, msg);
    }
    const uint32_t n_owned = (uint32_t)n_owned64;

    HIP_TRY(hipMemsetAsync(d_mean_out, 0, (size_t)3 * W * H * sizeof(double), stream));
    std::memset(&sc->prof, 0, sizeof(sc->prof));
    for (auto& e : sc->events) { sc->event_pool.push_back(e.second.first); sc->event_pool.push_back(e.second.second); }  // left by a call that failed
    sc->events.clear();
    if (n_owned == 0) { if (sync_at_end) HIP_TRY(hipStreamSynchronize(stream)); return hand_back(); }

    // Samples per pass: large passes amortise what every launch costs (the workgroups' copy of the node array into LDS,
    // the launch itself, the ragged end of a persistent kernel): 128 M paths a pass (120 bytes of workspace each, 16 GB) run
    // final_scene 1.5 % faster than 64 M, which ran
    // final_scene 16 % faster than 8 M.  Halved while the workspace does not fit.
    uint64_t S = ro->samples_per_pass;
    const bool auto_pass = S == 0;
    if (auto_pass) S = std::max<uint64_t>(1, (uint64_t)RT_AUTO_PASS_PATHS / n_owned);
    S = std::min<uint64_t>(S, spp);
    if (fixed) S = std::min<uint64_t>(S, fixed->paths / n_owned);  // (at least 1: n_owned <= pixels <= paths)
    while (S > 1 && S * n_owned >= (1ull << 31)) S /= 2;
    if (S * n_owned >= (1ull << 31)) return fail(RT_ERR_UNSUPPORTED, "too many pixels per rank for one pass");
    while (!fixed && (st = ensure_workspace(sc, S * n_owned, n_owned, n_counts)) != RT_OK) {
        if (st != RT_ERR_OUT_OF_MEMORY || !auto_pass || S == 1) return st;
        (void)hipGetLastError();  // the failed allocation is not an error of the render
        S = (S + 1) / 2;
    }
    const Workspace& ws = fixed ? *fixed : sc->ws;
    sc->prof.pass_paths = S * n_owned;
    {   // (n_owned != 0: the rank has at least one tile)
        const uint32_t my_tiles = (uint32_t)count_residue_below(tiles.ntiles(), tiles.rank, tiles.world);
        const uint32_t rows_per_block = std::max(16u, (tiles.th + 65534u) / 65535u);
        hipLaunchKernelGGL(k_owned_list, dim3(my_tiles, (tiles.th + rows_per_block - 1) / rows_per_block), dim3(RT_OWNED_BS), 0, stream, tiles, rows_per_block, ws.owned);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemsetAsync(ws.accum, 0, (size_t)n_owned * 3 * sizeof(double), stream));

    const uint32_t max_grid = sc->n_cu * 8u;
    size_t node_bytes = ((size_t)sc->view.n_bvh_nodes + 1) * sizeof(WalkNode);
    const bool nodes_in_lds = sc->nodes_fit_lds && !(ro->flags & RT_RENDER_NODES_IN_GLOBAL);
    size_t lds_bytes = node_bytes + RT_LDS_BEHIND_NODES + sc->view.lds_scene_bytes;  // dynamic LDS of the kernels that stage the nodes (changes when the tree is regrouped)
    const uint32_t tail_threshold = sc->n_cu * RT_TAIL_PER_CU;
    Launcher L{sc, stream, profile};
    hipEvent_t ev;

    PassParams P;
    P.cam = *cam; P.width = W; P.height = H; P.n_owned = n_owned;
    P.owned_magic = n_owned > 1 ? ~0ull / n_owned + 1 : 0;
    P.k0 = (uint32_t)ro->seed; P.k1 = (uint32_t)(ro->seed >> 32); P.max_depth = depth; P.owned = ws.owned;
    P.n_counts = n_counts; P.overlapped = 0;
    P.up_axis = 1;
    for (uint32_t c = 0; c < 3; ++c) if (std::fabs(cam->v[c]) > std::fabs(cam->v[P.up_axis])) P.up_axis = c;
    P.up_sign = cam->v[P.up_axis] < 0.0 ? -1.0f : 1.0f;

    const bool force_wavefront = (ro->flags & RT_RENDER_NO_TAIL) != 0, force_tail = (ro->flags & RT_RENDER_TAIL_ONLY) != 0;
    // one kernel per bounce (k_shade<true>: walk and shade) for trees of a few dozen node slots that fit in LDS
    const bool scan = sc->n_leaves <= RT_SCAN_MAX_LEAVES;
    const bool fused = nodes_in_lds && !(ro->flags & RT_RENDER_TWO_KERNELS) && ((ro->flags & RT_RENDER_ONE_KERNEL) || scan);
    int tail_from = force_tail ? 0 : -1;  // bounce at which k_tail takes over; -1 = not calibrated yet
    bool split_pays = false;               // set by the calibrating pass: traversal dominates shading
    Calibration* known = nullptr;
    for (auto& c : sc->calibrations)
        if (c.width == W && c.height == H && c.depth == depth && c.n_owned == n_owned && c.pass_paths == S * n_owned) known = &c;
    if (fixed && tail_from < 0 && !force_wavefront) {
        // nothing is read back: queues of a pass this small are short after a bounce or two
        tail_from = S * n_owned > tail_threshold ? (int)std::min<uint32_t>(depth, 2u) : 0;
    } else if (known && tail_from < 0 && !force_wavefront && !profile && !(ro->flags & RT_RENDER_RECALIBRATE)) {
        tail_from = known->tail_from; split_pays = known->split_pays;
    }

    // A pass runs as one or two independent halves (its samples split in two), each with its own queues and counters:
    // traversal is bound by instruction issue and shading by memory traffic, so the two halves, launched on two streams,
    // fill each other's gaps (the shading waves of one fit beside the traversal workgroup of the other on a CU).
    // Passes that read queue lengths back (the first of a new frame shape, profiled ones, RT_RENDER_NO_TAIL) run as one
    // half on one stream.
    struct Half {
        hipStream_t stream; PassParams P; RayQueue q[2]; uint2* hits; uint32_t* counts; float4* values; int cur;
    };
    const bool sides = sc->view.side_rounding != 0u;  // (which build of the shading kernels: see stage_nodes)
    const bool multi = sc->view.multi_scatter != 0u;  // (which build of the traversal kernel: see stage_nodes)
    const bool generic = sc->n_long_leaves != 0;       // (some leaf takes the general primitive test: the builds with that code)
    const auto traverse_lds = !generic ? &k_traverse<true, false, false> : (multi ? &k_traverse<true, true> : &k_traverse<true, false>);
    const auto traverse_global = !generic ? &k_traverse<false, false, false> : (multi ? &k_traverse<false, true> : &k_traverse<false, false>);
    const auto tail_lds = generic ? (sides ? &k_tail<true, true> : &k_tail<true, false>) : (sides ? &k_tail<true, true, false> : &k_tail<true, false, false>);
    const auto tail_global = generic ? (sides ? &k_tail<false, true> : &k_tail<false, false>) : (sides ? &k_tail<false, true, false> : &k_tail<false, false, false>);
    const auto shade_scan = sc->n_long_leaves ? (sides ? &k_shade<SHADE_SCAN, true, true> : &k_shade<SHADE_SCAN, false, true>)
                                              : (sides ? &k_shade<SHADE_SCAN, true, false> : &k_shade<SHADE_SCAN, false, false>);
    const auto shade_scan_first = sc->n_long_leaves ? (sides ? &k_shade<SHADE_SCAN, true, true, true> : &k_shade<SHADE_SCAN, false, true, true>)
                                                    : (sides ? &k_shade<SHADE_SCAN, true, false, true> : &k_shade<SHADE_SCAN, false, false, true>);
    const auto shade_scan_first_two = sides ? &k_shade<SHADE_SCAN, true, false, true, true> : &k_shade<SHADE_SCAN, false, false, true, true>;  // (scenes without long-test leaves only)
    const auto shade_scan_front = sides ? &k_shade<SHADE_SCAN, true, true, false, false, 1> : &k_shade<SHADE_SCAN, false, true, false, false, 1>;
    const auto shade_scan_back = sides ? &k_shade<SHADE_SCAN, true, true, false, false, 2> : &k_shade<SHADE_SCAN, false, true, false, false, 2>;
    const auto shade_walk = sides ? &k_shade<SHADE_WALK, true> : &k_shade<SHADE_WALK, false>;
    const auto shade_hits = sides ? &k_shade<SHADE_HITS, true> : &k_shade<SHADE_HITS, false>;
    auto launch_traverse = [&](Half& h, uint32_t b, uint32_t grid, uint32_t trav_grid) {
        if (nodes_in_lds)
            hipLaunchKernelGGL(traverse_lds, dim3(trav_grid), dim3(RT_TRAV_BS), lds_bytes, h.stream, sc->view, h.P, h.q[h.cur], h.counts + b, h.hits, b);
        else
            hipLaunchKernelGGL(traverse_global, dim3(grid), dim3(RT_BS), RT_LDS_BEHIND_NODES, h.stream, sc->view, h.P, h.q[h.cur], h.counts + b, h.hits, b);
    };
    auto launch_tail = [&](Half& h, uint32_t b, uint32_t grid, uint32_t trav_grid) {
        if (nodes_in_lds)
            hipLaunchKernelGGL(tail_lds, dim3(trav_grid), dim3(RT_TRAV_BS), lds_bytes, h.stream, sc->view, h.P, h.q[h.cur], h.counts + b,
                               h.values, h.counts + depth + 1, b);
        else
            hipLaunchKernelGGL(tail_global, dim3(grid), dim3(RT_BS), 0, h.stream, sc->view, h.P, h.q[h.cur], h.counts + b, h.values,
                               h.counts + depth + 1, b);
    };
    // max_reflections == 0: ray_color's loop body never runs and every sample is black (lib.rs:89, :103)
    for (uint64_t s0 = 0; s0 < spp && depth > 0; s0 += S) {
        const uint32_t ns = (uint32_t)std::min<uint64_t>(S, spp - s0);
        // Queue lengths are read back (one sync per bounce) only while the hand-over point to k_tail is
        // still unknown, when profiling, or when k_tail is disabled; otherwise nothing waits on the host.
        const bool readback = !fixed && (tail_from < 0 || profile || force_wavefront);
        // a scanned scene's first bounce makes its own camera rays (unless the pass goes to k_tail at once)
        const bool first_is_scan = RT_SCAN_FIRST && fused && scan && !(tail_from == 0 && !force_wavefront);
        // ... and its second bounce too, in the same lanes, when nothing has to be known about the queue between the two.  Nearly
        // every path has a second segment, so the lanes stay full, and 96 bytes per path stay out of memory (perlin, whose
        // paths are 2.3 segments long: 10 890 -> 11 950 Msamples/s).  Not where the queue is ordered by long-test leaves: the
        // second bounce would run unordered (cornell -2 %, cornell_smoke -9 %).
        const bool first_two = RT_SCAN_TWO && first_is_scan && !readback && depth >= 2 && tail_from >= 2 && sc->n_long_leaves == 0;
        const bool calibrating = !fixed && tail_from < 0 && !profile;  // this pass also times its kernels, to see whether splitting later passes pays
        L.profile = profile || calibrating;
        const int n_halves = (!fixed && !readback && split_pays && ns >= RT_PASS_PARTS && !(ro->flags & RT_RENDER_ONE_STREAM)) ? RT_PASS_PARTS : 1;
        Half half[RT_MAX_PARTS];
        uint32_t sample_off = 0;
        for (int k = 0; k < n_halves; ++k) {
            Half& h = half[k];
            const uint32_t ns_k = ns / n_halves + ((uint32_t)k < ns % n_halves ? 1u : 0u);
            const size_t path_off = (size_t)sample_off * n_owned;
            h.stream = k == 0 ? stream : sc->side_stream[k - 1];
            h.P = P; h.P.sample_base = (uint32_t)s0 + sample_off; h.P.n_paths = ns_k * n_owned; h.P.overlapped = n_halves > 1 ? 1u : 0u;
            for (int j = 0; j < 2; ++j) { h.q[j].od0 = ws.q[j].od0 + path_off; h.q[j].od1 = ws.q[j].od1 + path_off; h.q[j].thr = ws.q[j].thr + path_off; }
            h.hits = ws.hits + path_off; h.counts = ws.counts + (size_t)k * 2 * n_counts; h.values = ws.values + path_off; h.cur = 0;
            sample_off += ns_k;
        }
        if (n_halves > 1) {  // the other streams start after everything queued so far (the previous pass's resolve reads `values`)
            HIP_TRY(hipEventRecord(sc->ev_fork, stream));
            for (int k = 1; k < n_halves; ++k) HIP_TRY(hipStreamWaitEvent(sc->side_stream[k - 1], sc->ev_fork, 0));
        }
        uint32_t grid[RT_MAX_PARTS], shade_grid[RT_MAX_PARTS], scan_grid[RT_MAX_PARTS], trav_grid[RT_MAX_PARTS];
        for (int k = 0; k < n_halves; ++k) {
            Half& h = half[k];
            const uint32_t grid_full = (h.P.n_paths + RT_BS - 1) / RT_BS;
            grid[k] = std::min(grid_full, max_grid);
            shade_grid[k] = std::min((h.P.n_paths + RT_SHADE_BS - 1) / RT_SHADE_BS, max_grid * RT_BS / RT_SHADE_BS);
            scan_grid[k] = std::min((h.P.n_paths + shade_bs(SHADE_SCAN) - 1) / shade_bs(SHADE_SCAN), sc->n_cu * (sc->n_long_leaves ? RT_SCAN_GRID_PER_CU : RT_PLAIN_SCAN_GRID_PER_CU));
            trav_grid[k] = std::min((h.P.n_paths + RT_TRAV_BS - 1) / RT_TRAV_BS, sc->n_cu);
            if (k == 0 && (st = L.begin(RT_K_RAYGEN, &ev)) != RT_OK) return st;
            if (first_is_scan) hipLaunchKernelGGL(k_reset_counts, dim3(1), dim3(64), 0, h.stream, h.counts, n_counts, h.P.n_paths);
            else hipLaunchKernelGGL(k_raygen, dim3(grid_full), dim3(RT_BS), 0, h.stream, h.P, h.q[0], h.counts, n_counts);
            if (k == 0 && (st = L.end(ev)) != RT_OK) return st;
            sc->prof.samples += h.P.n_paths;
        }
        sc->prof.passes += 1;
        sc->prof.two_stream_passes += n_halves > 1 ? 1u : 0u;
        sc->prof.two_bounce_passes += first_two ? 1u : 0u;
        // The first sizeable pass on a scene handle lends some of its rays to the regrouping of the tree (tune_tree).
        const bool sampling = !fixed && !sc->tree_tuned && !(ro->flags & RT_RENDER_KEEP_TREE) && !fused && sc->n_leaves > RT_SCAN_MAX_LEAVES &&
                              half[0].P.n_paths >= RT_TUNE_MIN_PATHS;
        uint32_t sampled_bounces = 0;
        uint32_t b = 0;
        bool finished = false;
        for (; b < depth; ++b) {
            if (tail_from >= 0 && !force_wavefront && (int)b >= tail_from) break;
            const bool two = b == 0 && first_two;  // this launch runs bounces 0 and 1
            for (int k = 0; k < n_halves; ++k) {  // (event timing only in one-half passes: L works on `stream`)
                Half& h = half[k];
                if (sampling && k == 0 && b < RT_TUNE_BOUNCES) {
                    hipLaunchKernelGGL(k_sample_rays, dim3(1), dim3(64), 0, h.stream, h.q[h.cur], h.counts + b, n_counts, h.P.n_paths, RT_TUNE_PER_BOUNCE,
                                       sc->d_tune_rays + (size_t)b * RT_TUNE_PER_BOUNCE * 6);
                    sampled_bounces = b + 1;
                }
                if (!fused) {
                    if (n_halves == 1 && (st = L.begin(RT_K_TRAVERSE, &ev)) != RT_OK) return st;
                    launch_traverse(h, b, grid[k], trav_grid[k]);
                    if (n_halves == 1 && (st = L.end(ev)) != RT_OK) return st;
                }
                if (n_halves == 1 && (st = L.begin(RT_K_SHADE, &ev)) != RT_OK) return st;
                if (fused && scan && RT_SCAN_SPLIT && sc->n_long_leaves && !(b == 0 && first_is_scan)) {
                    // the queue is ordered by long-test leaves: its back part runs the build without the general test
                    hipLaunchKernelGGL(shade_scan_front, dim3(scan_grid[k]), dim3(shade_bs(SHADE_SCAN)), lds_bytes, h.stream, sc->view, h.P, h.q[h.cur], h.hits, h.q[h.cur ^ 1],
                                       h.counts + b, h.counts + b + 1, h.values, b);
                    hipLaunchKernelGGL(shade_scan_back, dim3(scan_grid[k]), dim3(shade_bs(SHADE_SCAN)), lds_bytes, h.stream, sc->view, h.P, h.q[h.cur], h.hits, h.q[h.cur ^ 1],
                                       h.counts + b, h.counts + b + 1, h.values, b);
                } else if (fused && scan)
                    hipLaunchKernelGGL(two ? shade_scan_first_two : (b == 0 && first_is_scan ? shade_scan_first : shade_scan), dim3(scan_grid[k]), dim3(shade_bs(SHADE_SCAN)), lds_bytes,
                                       h.stream, sc->view, h.P, h.q[h.cur], h.hits, h.q[h.cur ^ 1], h.counts + b, h.counts + b + (two ? 2 : 1), h.values, b);
                else if (fused)
                    hipLaunchKernelGGL(shade_walk, dim3(shade_grid[k]), dim3(RT_SHADE_BS), lds_bytes, h.stream, sc->view, h.P,
                                       h.q[h.cur], h.hits, h.q[h.cur ^ 1], h.counts + b, h.counts + b + 1, h.values, b);
                else
                    hipLaunchKernelGGL(shade_hits, dim3(shade_grid[k]), dim3(RT_SHADE_BS), 0, h.stream, sc->view, h.P,
                                       h.q[h.cur], h.hits, h.q[h.cur ^ 1], h.counts + b, h.counts + b + 1, h.values, b);
                if (n_halves == 1 && (st = L.end(ev)) != RT_OK) return st;
                h.cur ^= 1;
            }
            HIP_TRY(hipGetLastError());
            if (two) ++b;
            if (readback) {
                HIP_TRY(hipMemcpyAsync(sc->h_counts, half[0].counts + b, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
                HIP_TRY(hipMemcpyAsync(sc->h_counts + 2, half[0].counts + n_counts + b, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
                HIP_TRY(hipStreamSynchronize(stream));
                const uint32_t now = sc->h_counts[0] + sc->h_counts[2], next = sc->h_counts[1] + sc->h_counts[3];
                sc->prof.segments += now; sc->prof.traverse_segments += now;
                if (next == 0) { finished = true; break; }
                if (tail_from < 0 && !force_wavefront && next <= tail_threshold) tail_from = (int)b + 1;
            }
        }
        if (tail_from < 0 && !force_wavefront) tail_from = (int)depth;
        sc->prof.tail_from = tail_from >= 0 ? tail_from : (int32_t)depth;
        if (!finished && b < depth) {
            for (int k = 0; k < n_halves; ++k) {
                if (n_halves == 1 && (st = L.begin(RT_K_TAIL, &ev)) != RT_OK) return st;
                launch_tail(half[k], b, grid[k], trav_grid[k]);
                if (n_halves == 1 && (st = L.end(ev)) != RT_OK) return st;
            }
            HIP_TRY(hipGetLastError());
            if (profile) {
                HIP_TRY(hipMemcpyAsync(sc->h_counts, half[0].counts + depth + 1, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
                HIP_TRY(hipStreamSynchronize(stream));
                sc->prof.segments += sc->h_counts[0];
            }
        }
        for (int k = 1; k < n_halves; ++k) {  // every part's values are complete before the per-pixel sums
            HIP_TRY(hipEventRecord(sc->ev_join[k - 1], sc->side_stream[k - 1]));
            HIP_TRY(hipStreamWaitEvent(stream, sc->ev_join[k - 1], 0));
        }
        if ((st = L.begin(RT_K_RESOLVE, &ev)) != RT_OK) return st;
        hipLaunchKernelGGL(k_resolve, dim3((n_owned + RT_BS - 1) / RT_BS), dim3(RT_BS), 0, stream, ws.values, ws.accum, n_owned, ns);
        if ((st = L.end(ev)) != RT_OK) return st;
        if (calibrating) {
            // Two halves on two streams pay when the instruction-bound traversal dominates the memory-bound shading (they
            // then fill each other's gaps), and when a bounce is one kernel (the last workgroups of one half's launch run
            // beside the first of the other's: 5 % on the Cornell rooms, three or four parts are slower again).  Two-kernel
            // bounces of small trees keep one stream: both kernels wait on the same queue traffic.
            HIP_TRY(hipStreamSynchronize(stream));
            const rt_profile keep = sc->prof;
            if ((st = collect_profile(sc)) != RT_OK) return st;
            split_pays = fused || sc->prof.kernel_ms[RT_K_TRAVERSE] - keep.kernel_ms[RT_K_TRAVERSE] > 1.5 * (sc->prof.kernel_ms[RT_K_SHADE] - keep.kernel_ms[RT_K_SHADE]);
            sc->prof = keep;
            if (!force_wavefront && !force_tail && ns == S) {  // (a full-size pass: what it saw holds for the passes to come)
                const Calibration c{W, H, depth, n_owned, S * n_owned, tail_from, split_pays};
                if (known) *known = c;
                else { if (sc->calibrations.size() >= 64) sc->calibrations.erase(sc->calibrations.begin()); sc->calibrations.push_back(c); known = &sc->calibrations.back(); }
            }
        }
        if (sampling && sampled_bounces > 0) {
            HIP_TRY(hipStreamSynchronize(stream));  // (the pass is complete: resolve has waited for every part)
            if ((st = tune_tree(sc, sampled_bounces)) != RT_OK) return st;
            node_bytes = ((size_t)sc->view.n_bvh_nodes + 1) * sizeof(WalkNode);
            lds_bytes = node_bytes + RT_LDS_BEHIND_NODES + sc->view.lds_scene_bytes;
        }
    }
    L.profile = profile;
    if ((st = L.begin(RT_K_FINALIZE, &ev)) != RT_OK) return st;
    hipLaunchKernelGGL(k_finalize, dim3((n_owned + RT_BS - 1) / RT_BS), dim3(RT_BS), 0, stream, ws.accum, ws.owned, n_owned, W, H, (double)spp, d_mean_out);
    if ((st = L.end(ev)) != RT_OK) return st;
    if (sync_at_end || profile) {
        HIP_TRY(hipStreamSynchronize(stream));
        if (profile && (st = collect_profile(sc)) != RT_OK) return st;
    }
    return hand_back();
}

static rt_status quantise_impl(rt_scene* sc, const double* d_mean, uint64_t n, uint8_t* d_rgb_out, void* stream);

extern "C" {

rt_status rt_render_device(rt_scene* sc, const rt_camera* cam, const rt_image_options* io, const rt_render_options* ro, double* d_mean_out) {
    if (!sc) return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
    std::lock_guard<std::mutex> lock(sc->mutex);
    return render_impl(sc, cam, io, ro, d_mean_out, false, nullptr);
}

rt_status rt_render_device_noalloc(rt_scene* sc, const rt_camera* cam, const rt_image_options* io, const rt_render_options* ro, double* d_mean_out) {
    if (!sc) return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
    unsigned char* base = (sc->device >= 0 && sc->device < RT_MAX_DEVICES) ? g_scratch_ptr[sc->device] : nullptr;
    if (!base) return fail(RT_ERR_UNSUPPORTED, "this device has no static workspace (device index beyond the library's table)");
    std::lock_guard<std::mutex> lock(sc->mutex);
    const Workspace ws = static_workspace(base);
    return render_impl(sc, cam, io, ro, d_mean_out, false, &ws);
}

rt_status rt_quantise_device(rt_scene* sc, const double* d_mean, uint64_t n, uint8_t* d_rgb_out, void* stream) {
    if (!sc || !d_mean || !d_rgb_out) return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
    std::lock_guard<std::mutex> lock(sc->mutex);
    return quantise_impl(sc, d_mean, n, d_rgb_out, stream);
}
}  // extern "C"
static rt_status quantise_impl(rt_scene* sc, const double* d_mean, uint64_t n, uint8_t* d_rgb_out, void* stream) {
    HIP_TRY(hipSetDevice(sc->device));
    hipStream_t s = stream ? (hipStream_t)stream : sc->own_stream;
    if (n == 0) return RT_OK;
    hipLaunchKernelGGL(k_quantise, dim3((uint32_t)((n + RT_BS - 1) / RT_BS)), dim3(RT_BS), 0, s, d_mean, n, d_rgb_out);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}
extern "C" {

rt_status rt_render(rt_scene* sc, const rt_camera* cam, const rt_image_options* io, const rt_render_options* ro, uint8_t* rgb_out, double* mean_out) {
    rt_status st = validate_render(sc, cam, io, ro);
    if (st != RT_OK) return st;
    if (!rgb_out && !mean_out) return fail(RT_ERR_INVALID_ARGUMENT, "no output buffer");
    std::lock_guard<std::mutex> lock(sc->mutex);
    HIP_TRY(hipSetDevice(sc->device));
    const size_t n = (size_t)3 * io->width * io->height;
    double* d_mean = nullptr; uint8_t* d_rgb = nullptr;
    HIP_TRY(hipMalloc((void**)&d_mean, n * sizeof(double)));
    hipStream_t stream = ro->stream ? (hipStream_t)ro->stream : sc->own_stream;
    st = render_impl(sc, cam, io, ro, d_mean, false, nullptr);
    if (st == RT_OK && rgb_out) {
        hipError_t e = hipMalloc((void**)&d_rgb, n);
        if (e != hipSuccess) st = fail(RT_ERR_OUT_OF_MEMORY, "hipMalloc(rgb)");
        if (st == RT_OK) st = quantise_impl(sc, d_mean, n, d_rgb, stream);
        if (st == RT_OK && hipMemcpyAsync(rgb_out, d_rgb, n, hipMemcpyDeviceToHost, stream) != hipSuccess) st = fail(RT_ERR_HIP, "copy of the image to the host failed");
    }
    if (st == RT_OK && mean_out && hipMemcpyAsync(mean_out, d_mean, n * sizeof(double), hipMemcpyDeviceToHost, stream) != hipSuccess)
        st = fail(RT_ERR_HIP, "copy of the mean radiance to the host failed");
    hipError_t e = hipStreamSynchronize(stream);
    if (st == RT_OK && e != hipSuccess) st = fail(RT_ERR_HIP, std::string("render failed: ") + hipGetErrorString(e));
    if (st == RT_OK && (ro->flags & RT_RENDER_PROFILE)) st = collect_profile(sc);
    (void)hipFree(d_mean);
    if (d_rgb) (void)hipFree(d_rgb);
    return st;
}

#ifdef RT_TRAV_STATS
// Analysis build only: reads and clears the turn statistics of k_traverse.
rt_status rt_debug_trav_stats(unsigned long long* out16) {
    unsigned long long zero[16] = {};
    HIP_TRY(hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_trav_stats), sizeof(zero)));
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_trav_stats), zero, sizeof(zero)));
    return RT_OK;
}
#endif

#ifdef RT_SHADE_STATS
// Analysis build only: reads and clears the wave statistics of k_shade.
rt_status rt_debug_shade_stats(unsigned long long* out40) {
    unsigned long long zero[40] = {};
    HIP_TRY(hipMemcpyFromSymbol(out40, HIP_SYMBOL(g_shade_stats), sizeof(zero)));
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_shade_stats), zero, sizeof(zero)));
    return RT_OK;
}
#endif

rt_status rt_get_profile(rt_scene* sc, rt_profile* out) {
    if (!sc || !out) return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
    std::lock_guard<std::mutex> lock(sc->mutex);
    *out = sc->prof;
    out->tree_tests_by_area = sc->tree_tests_by_area; out->tree_tests_sampled = sc->tree_tests_sampled;  // (of the handle, not of the last call)
    return RT_OK;
}

rt_status rt_intersect_batch(rt_scene* sc, int64_t scope, const rt_ray* rays, uint64_t n, float tmin, float tmax, uint64_t seed, const float* medium_u, rt_hit* out) {
    if (!sc || (!rays && n) || (!out && n)) return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
    if (scope >= (int64_t)sc->view.n_prims) return fail(RT_ERR_INVALID_ARGUMENT, "scope is not a primitive of this scene");
    if (n == 0) return RT_OK;
    if (n >= (1ull << 31)) return fail(RT_ERR_UNSUPPORTED, "batch too large");
    std::lock_guard<std::mutex> lock(sc->mutex);
    HIP_TRY(hipSetDevice(sc->device));
    rt_ray* d_rays = nullptr; rt_hit* d_out = nullptr; float* d_mu = nullptr;
    rt_status st = RT_OK;
    auto cleanup = [&]() { if (d_rays) (void)hipFree(d_rays); if (d_out) (void)hipFree(d_out); if (d_mu) (void)hipFree(d_mu); };
#define TRY2(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { cleanup(); return fail(RT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } } while (0)
    TRY2(hipMalloc((void**)&d_rays, n * sizeof(rt_ray)));
    TRY2(hipMalloc((void**)&d_out, n * sizeof(rt_hit)));
    TRY2(hipMemcpy(d_rays, rays, n * sizeof(rt_ray), hipMemcpyHostToDevice));
    if (medium_u) { TRY2(hipMalloc((void**)&d_mu, n * sizeof(float))); TRY2(hipMemcpy(d_mu, medium_u, n * sizeof(float), hipMemcpyHostToDevice)); }
    hipLaunchKernelGGL(k_intersect_batch, dim3((uint32_t)((n + RT_BS - 1) / RT_BS)), dim3(RT_BS), 0, sc->own_stream, sc->view, scope, d_rays, n, tmin,
                       tmax, (uint32_t)seed, (uint32_t)(seed >> 32), d_mu, d_out);
    TRY2(hipGetLastError());
    TRY2(hipStreamSynchronize(sc->own_stream));
    TRY2(hipMemcpy(out, d_out, n * sizeof(rt_hit), hipMemcpyDeviceToHost));
    cleanup();
    return st;
}

rt_status rt_scatter_batch(rt_scene* sc, const rt_hit* hits, uint64_t n, uint64_t seed, const uint32_t* keys, rt_scatter* out) {
    if (!sc || (!hits && n) || (!out && n)) return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
    if (n == 0) return RT_OK;
    if (n >= (1ull << 31)) return fail(RT_ERR_UNSUPPORTED, "batch too large");
    for (uint64_t i = 0; i < n; ++i)
        if (hits[i].prim >= sc->view.n_prims) return fail(RT_ERR_INVALID_ARGUMENT, "hit record names a primitive this scene does not have");
    std::lock_guard<std::mutex> lock(sc->mutex);
    HIP_TRY(hipSetDevice(sc->device));
    rt_hit* d_hits = nullptr; rt_scatter* d_out = nullptr; uint32_t* d_keys = nullptr;
    auto cleanup = [&]() { if (d_hits) (void)hipFree(d_hits); if (d_out) (void)hipFree(d_out); if (d_keys) (void)hipFree(d_keys); };
    TRY2(hipMalloc((void**)&d_hits, n * sizeof(rt_hit)));
    TRY2(hipMalloc((void**)&d_out, n * sizeof(rt_scatter)));
    TRY2(hipMemcpy(d_hits, hits, n * sizeof(rt_hit), hipMemcpyHostToDevice));
    if (keys) { TRY2(hipMalloc((void**)&d_keys, 3 * n * sizeof(uint32_t))); TRY2(hipMemcpy(d_keys, keys, 3 * n * sizeof(uint32_t), hipMemcpyHostToDevice)); }
    hipLaunchKernelGGL(k_scatter_batch, dim3((uint32_t)((n + RT_BS - 1) / RT_BS)), dim3(RT_BS), 0, sc->own_stream, sc->view, d_hits, n, (uint32_t)seed,
                       (uint32_t)(seed >> 32), d_keys, d_out);
    TRY2(hipGetLastError());
    TRY2(hipStreamSynchronize(sc->own_stream));
    TRY2(hipMemcpy(out, d_out, n * sizeof(rt_scatter), hipMemcpyDeviceToHost));
    cleanup();
    return RT_OK;
}

rt_status rt_camera_rays(rt_scene* sc, const rt_camera* cam, const rt_image_options* io, uint64_t seed, uint64_t n, const uint32_t* px_i,
                         const uint32_t* px_j, const uint32_t* sample, rt_ray* out) {
    if (!sc || !cam || !io || ((!px_i || !px_j || !sample || !out) && n)) return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
    if (io->width < 2 || io->height < 2) return fail(RT_ERR_INVALID_ARGUMENT, "width and height must be at least 2");
    if (n == 0) return RT_OK;
    for (uint64_t k = 0; k < n; ++k)
        if (px_i[k] >= io->width || px_j[k] >= io->height) return fail(RT_ERR_INVALID_ARGUMENT, "pixel out of range");
    std::lock_guard<std::mutex> lock(sc->mutex);
    HIP_TRY(hipSetDevice(sc->device));
    uint32_t* d_idx = nullptr; rt_ray* d_out = nullptr;
    auto cleanup = [&]() { if (d_idx) (void)hipFree(d_idx); if (d_out) (void)hipFree(d_out); };
    TRY2(hipMalloc((void**)&d_idx, 3 * n * sizeof(uint32_t)));
    TRY2(hipMalloc((void**)&d_out, n * sizeof(rt_ray)));
    TRY2(hipMemcpy(d_idx, px_i, n * sizeof(uint32_t), hipMemcpyHostToDevice));
    TRY2(hipMemcpy(d_idx + n, px_j, n * sizeof(uint32_t), hipMemcpyHostToDevice));
    TRY2(hipMemcpy(d_idx + 2 * n, sample, n * sizeof(uint32_t), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_camera_rays, dim3((uint32_t)((n + RT_BS - 1) / RT_BS)), dim3(RT_BS), 0, sc->own_stream, *cam, io->width, io->height, (uint32_t)seed,
                       (uint32_t)(seed >> 32), n, d_idx, d_idx + n, d_idx + 2 * n, d_out);
    TRY2(hipGetLastError());
    TRY2(hipStreamSynchronize(sc->own_stream));
    TRY2(hipMemcpy(out, d_out, n * sizeof(rt_ray), hipMemcpyDeviceToHost));
    cleanup();
    return RT_OK;
#undef TRY2
}

}  // extern "C"
