// rt_bvh.hpp — bounding volume hierarchy over the triangles: build kernels and the per-lane
// traversal kernels (gfx950, wave64).
//
// The hierarchy only decides which EXACT tests are evaluated; every reported hit comes from
// the same exact arithmetic as the linear scans, and the winner rule of the scans is kept
// explicitly: smallest t, then smallest triangle index (ray_tracer.cpp:214,
// opencl_executor.cpp:242).  Boxes are inflated per ray by more than the distance at which the
// exact tests can still accept a hit outside its triangle (see TRay), so a subtree is skipped
// only if no exact test in it can succeed.
//
// Build (all on the device, once per scene):
//   k_bvh_keys    30-bit Morton code of the centre of each triangle's box, key = code << 32 | triangle index
//   radix sort    hipcub::DeviceRadixSort on the 64-bit keys (unique by construction)
//   k_bvh_leaves  triangle records gathered into key order, leaf boxes
//   k_bvh_split   binary radix tree over the sorted keys, one thread per internal node
//                 (longest-common-prefix construction; every node knows its key range)
//   k_bvh_refit   boxes bottom-up inside blocks of 256 leaves, second thread to arrive at a node computes it;
//   k_bvh_refit_top  the nodes that reach across those blocks, one block, round by round
//   k_bvh_emit    4-wide traversal nodes: the boxes of a node's grandchildren and the node's normal
//                 cone in one 128-byte record (a cache line), so a ray makes half the dependent fetches of a binary
//                 walk; subtrees with at most BVH_LEAF_MAX triangles become leaves (contiguous in key order)
// Frame kernels: k_bvh_closest_pk (makes the camera rays and walks the 64 of a wave as one packet), k_bvh_closest (bounce
// rays and batch calls, one ray per lane; k_bvh_closest_p is its form whose lanes take a new ray when theirs is decided,
// kept for A/B runs), k_bvh_shadow_direct (a lane keeps its sample and traces its shadow rays lamp after lamp, then
// shades the sample; k_bvh_shadow is the older form with a per-block list of (sample, lamp) pairs), k_bvh_tail (follows
// the rays of a small queue to the end of their paths), k_bvh_anyhit (batch calls).  The per-lane kernels are templates
// on the block size: launches of few rays use one-wave blocks.
#pragma once
#include "rt_kernels.hpp"

namespace rt {

constexpr uint32_t BVH_LEAF = 0x80000000u;      // reference is a leaf: bits 0..27 first triangle, 28..30 count
constexpr uint32_t BVH_NONE = 0x7fffffffu;      // no node
#ifndef RT_BVH_LEAF_MAX
#define RT_BVH_LEAF_MAX 2                       // triangles a leaf holds at most (3 bits in a reference: up to 7)
#endif
constexpr int BVH_LEAF_MAX = RT_BVH_LEAF_MAX;
#ifndef RT_BVH_STACK
#define RT_BVH_STACK 21
#endif
#ifndef RT_CLOSEST_WAVES
#define RT_CLOSEST_WAVES 7
#endif
#ifndef RT_SCHED_W
#define RT_SCHED_W 16                           // step scheduling weight of the per-lane walk, see bvh_traverse
#endif
#ifndef RT_SHADOW_WAVES
#define RT_SHADOW_WAVES 5
#endif
#ifndef RT_BOX_ONE_CMP
#define RT_BOX_ONE_CMP 1                        // 0: the box test decides with three comparisons, see bvh_box
#endif
#ifndef RT_SHADOW_WAVES_MANY
#define RT_SHADOW_WAVES_MANY 7                  // waves per SIMD the shadow kernel is compiled for from two lamps up, see k_bvh_shadow_direct
#endif
#ifndef RT_BVH_SPLIT_LEAVES
#define RT_BVH_SPLIT_LEAVES 0                   // 1: free slots of a node are filled by opening its two-triangle leaves, see k_bvh_emit
#endif
constexpr bool BVH_SPLIT_LEAVES = RT_BVH_SPLIT_LEAVES != 0;
constexpr int BVH_STACK = RT_BVH_STACK;         // LDS stack entries per lane (a node pushes up to three)
constexpr uint32_t BVH_MAX_TRIS = 1u << 28;
constexpr float BVH_LOOSE = 3.0e-2f;            // see TRay
constexpr float BVH_TIGHT = 4.0e-6f;
constexpr float BVH_TIGHT_MIN = 500.0f;         // tight bound used only if (|cos| * sine) >= BVH_TIGHT * BVH_TIGHT_MIN = 2e-3
constexpr int BVH_NODE_F4 = 8;                  // float4 per inner-node record: 128 bytes, one cache line

struct DBvh {
    const float4 *nodes;    // BVH_NODE_F4 float4 per internal node, 128-byte aligned: 4 x {centre, w}, 4 x {half extent, ref}
                            // of the boxes of up to four grandchildren (unused slots have ref BVH_NONE; centre +- half
                            // extent contains the box).  The four w words carry what belongs to the node as a whole:
                            // w0 = two bfloat16 {A (high half), sine * k (low half)}, both rounded up, w1..w3 = axis * k
                            // of the cone of the node's own triangles
    const float4 *tris;     // 3 float4 per triangle in key order: {p, c21.x} {e1, c21.y} {e2, c21.z}
    const uint32_t *ids;    // triangle index of the k-th record
    uint32_t root;          // reference of the root (BVH_NONE for an empty scene)
    uint32_t n;             // triangles
    float scene_mag;        // bound on |coordinate| of any box corner
    float inv_clamp;        // largest |1 / direction component| the slab test uses (see tray_make)
    int stack_limit;        // usable entries of the traversal stack, <= BVH_STACK (lowered only to test the overflow path)
};

// ---------------------------------------------------------------------------------------
// build
// ---------------------------------------------------------------------------------------
DEV uint32_t morton_spread(uint32_t v) {        // 10 bits -> every third bit
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

DEV uint32_t morton_spread2(uint32_t v) {       // 10 bits -> every second bit
    v &= 0x3ffu;
    v = (v | (v << 8)) & 0x00ff00ffu;
    v = (v | (v << 4)) & 0x0f0f0f0fu;
    v = (v | (v << 2)) & 0x33333333u;
    v = (v | (v << 1)) & 0x55555555u;
    return v;
}

// `minor` = -1: the three axes interleaved; 0..2: that axis' ten bits come last, the other two are interleaved above them;
// `classes`: leading bits name the orientation class (the code loses as many of its last bits), see below
// Orientation class of a triangle for the leading code bits: 0 for a triangle too small for the loose enlargement of the
// boxes to matter (|e1||e2| <= big_thr) and not part of a finely tiled plane, else 1 + the cell of its unit normal (one hemisphere, as the cones take it) in a
// 63 x 63 octahedral map, the cell's two indices bit-interleaved so that neighbouring directions share leading bits; the axis
// directions fall in the middle of cells.  (rt_bvh_host.hpp has the same function for the host build.)
#ifndef RT_BVH_CELL_BITS
#define RT_BVH_CELL_BITS 6
#endif
constexpr int BVH_CELL_BITS = RT_BVH_CELL_BITS;      // bits per index of the normal map: 63 x 63 cells of about 3 degrees (4: 15 x 15, 5: 31 x 31;
                                                     // boxes at random angles: 68 / 47 / 37 ms per frame with 4 / 5 / 6)
DEV uint32_t bvh_normal_class(f3 c, f3 e1, f3 e2, float big_thr, const uint32_t *__restrict__ popular) {
    if (c.x * 0.267f + c.y * 0.535f + c.z * 0.802f < 0.f) c = -c;
    const float l1 = fabsf(c.x) + fabsf(c.y) + fabsf(c.z);
    if (!(l1 > 0.f) || !(l1 < 3.0e38f)) return 0u;
    float u = c.x / l1, v = c.y / l1;
    if (c.z < 0.f) { const float uu = (1.0f - fabsf(v)) * (u < 0.f ? -1.0f : 1.0f), vv = (1.0f - fabsf(u)) * (v < 0.f ? -1.0f : 1.0f); u = uu; v = vv; }
    const float half = (float)((1 << BVH_CELL_BITS) / 2 - 1);      // cells per half axis
    const uint32_t qu = (uint32_t)fminf(2.0f * half, fmaxf(0.0f, (u + 1.0f) * half + 0.5f)), qv = (uint32_t)fminf(2.0f * half, fmaxf(0.0f, (v + 1.0f) * half + 0.5f));
    uint32_t bin = 0;
    for (int b = 0; b < BVH_CELL_BITS; b++) bin |= (((qu >> b) & 1u) << (2 * b)) | (((qv >> b) & 1u) << (2 * b + 1));
    // large, or in a cell that holds a fifth of the scene's triangle area (a finely tiled plane: rt_bvh_host.hpp, popular_cells)
    const float ee = sqrtf(dot(e1, e1)) * sqrtf(dot(e2, e2));
    if (!(ee > big_thr) && !(popular && ((popular[bin >> 5] >> (bin & 31u)) & 1u))) return 0u;
    return 1u + bin;
}

__global__ void k_bvh_keys(uint32_t n, const float4 *__restrict__ tri_exact, f3 lo, f3 inv_ext, int minor, int classes, float big_thr,
                           const uint32_t *__restrict__ popular, unsigned long long *__restrict__ keys)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    f3 p = xyz(tri_exact[3 * (size_t)i]), e1 = xyz(tri_exact[3 * (size_t)i + 1]), e2 = xyz(tri_exact[3 * (size_t)i + 2]);
    // Centre of the triangle's BOX, not of its vertices: the two triangles a quad was cut into share it, get the same code, sort
    // next to each other and become one two-triangle leaf with the quad's box (by their vertex centroids they straddle grid
    // planes, and a leaf then pairs halves of neighbouring quads).  Node visits per ray -10 %, exact tests -25 %; frames with
    // this build: C2 2.73 -> 2.37 ms, C4 1.04 -> 0.92 ms, C5 16.5 -> 14.6 ms, C1 0.239 -> 0.218 ms.
    f3 a = p + e1, b = p + e2;
    f3 c = mk(0.5f * (fminf(p.x, fminf(a.x, b.x)) + fmaxf(p.x, fmaxf(a.x, b.x))), 0.5f * (fminf(p.y, fminf(a.y, b.y)) + fmaxf(p.y, fmaxf(a.y, b.y))),
              0.5f * (fminf(p.z, fminf(a.z, b.z)) + fmaxf(p.z, fmaxf(a.z, b.z))));
    float q[3] = {(c.x - lo.x) * inv_ext.x, (c.y - lo.y) * inv_ext.y, (c.z - lo.z) * inv_ext.z};
    uint32_t g[3];
    for (int k = 0; k < 3; k++) {
        float f = q[k] * 1024.0f;
        g[k] = (f > 0.f) ? (f < 1023.0f ? (uint32_t)f : 1023u) : 0u;        // NaN -> 0
    }
    uint32_t code = morton_spread(g[0]) | (morton_spread(g[1]) << 1) | (morton_spread(g[2]) << 2);
    const uint32_t code3 = code;
    if (minor >= 0) {
        const uint32_t ga = g[minor == 0 ? 1 : 0], gb = g[minor == 2 ? 1 : 2];
        code = ((morton_spread2(ga) | (morton_spread2(gb) << 1)) << 10) | g[minor];
    }
    // Orientation classes lead the code, so that they become subtrees under the root.  A node's normal cone decides whether a
    // ray may use the tight enlargement of its boxes (see TRay); in a subtree that mixes walls, floors and roofs no cone exists,
    // every box takes the loose enlargement (proportional to |e1||e2| and the distance), and nothing is culled: a scene of
    // 1500 boxes on a ground grid took 288 ms per 720p frame, 11.7 ms with the classes apart; C1 (a box on a plane) 0.212 ->
    // 0.110 ms; the terrains, whose triangles are small, are unchanged.  classes == 1: the axis the normal is closest to
    // (two bits); classes == 2, the default: for large triangles the cell of the normal in an octahedral map (walls that
    // stand at any angle: 150 -> 37 ms for the same boxes turned by random angles), small triangles stay together.
    if (classes == 1) {
        const float nx = fabsf(tri_exact[3 * (size_t)i].w), ny = fabsf(tri_exact[3 * (size_t)i + 1].w), nz = fabsf(tri_exact[3 * (size_t)i + 2].w);
        const uint32_t cls = (nx >= ny && nx >= nz) ? 0u : (ny >= nz ? 1u : 2u);
        code = (code >> 2) | (cls << 28);
    } else if (classes == 2) {
        const uint32_t cls = bvh_normal_class(mk(tri_exact[3 * (size_t)i].w, tri_exact[3 * (size_t)i + 1].w, tri_exact[3 * (size_t)i + 2].w), e1, e2, big_thr, popular);
        if (cls) code = (1u << 30) | ((cls - 1u) << (30 - 2 * BVH_CELL_BITS)) | (code3 >> (2 * BVH_CELL_BITS));        // (all three axes interleaved)
    }
    keys[i] = ((unsigned long long)code << 32) | i;
}

// build-time node numbering: internal nodes 0 .. n-2, leaf k is node (n-1) + k
__global__ void k_bvh_leaves(uint32_t n, const unsigned long long *__restrict__ keys, const float4 *__restrict__ tri_exact,
                             float4 *__restrict__ tris, uint32_t *__restrict__ ids, float4 *__restrict__ box_lo,
                             float4 *__restrict__ box_hi, float4 *__restrict__ nrm_lo, float4 *__restrict__ nrm_hi)
{
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    uint32_t i = (uint32_t)(keys[k] & 0xffffffffull);
    float4 r0 = tri_exact[3 * (size_t)i], r1 = tri_exact[3 * (size_t)i + 1], r2 = tri_exact[3 * (size_t)i + 2];
    tris[3 * (size_t)k] = r0; tris[3 * (size_t)k + 1] = r1; tris[3 * (size_t)k + 2] = r2;
    ids[k] = i;
    f3 p = xyz(r0), a = p + xyz(r1), b = p + xyz(r2);
    f3 e1 = xyz(r1), e2 = xyz(r2);
    float ee = sqrtf(dot(e1, e1)) * sqrtf(dot(e2, e2));
    float A = BVH_LOOSE * ee;
    if (!(A < 3.0e38f)) A = 3.0e38f;
    size_t node = (size_t)(n - 1) + k;
    // unit normal (either sign: only |dot(d, n)| matters; one hemisphere is chosen so that the
    // normals of a smooth patch stay together) and the sine of the angle between the edges
    f3 c21 = mk(r0.w, r1.w, r2.w);
    float cl = sqrtf(dot(c21, c21));
    float sine = cl / ee;
    f3 nn = c21 * (1.0f / cl);
    if (!(cl > 0.f) || !(sine > 0.f) || !(sine < 2.f) || !(fabsf(nn.x) <= 1.f) || !(fabsf(nn.y) <= 1.f) || !(fabsf(nn.z) <= 1.f)) {
        sine = 0.f; nn = mk(0.f, 0.f, 0.f);            // degenerate: never hit, and no tight bound for its ancestors
    }
    if (nn.x * 0.267f + nn.y * 0.535f + nn.z * 0.802f < 0.f) nn = -nn;
    nrm_lo[node] = make_float4(nn.x, nn.y, nn.z, fminf(sine, 1.0f));
    nrm_hi[node] = make_float4(nn.x, nn.y, nn.z, 0.f);
    box_lo[node] = make_float4(fminf(p.x, fminf(a.x, b.x)), fminf(p.y, fminf(a.y, b.y)), fminf(p.z, fminf(a.z, b.z)), A);
    box_hi[node] = make_float4(fmaxf(p.x, fmaxf(a.x, b.x)), fmaxf(p.y, fmaxf(a.y, b.y)), fmaxf(p.z, fmaxf(a.z, b.z)), 0.f);
}

// length of the common prefix of keys i and j, -1 outside [0, n)
DEV int bvh_delta(const unsigned long long *keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    return __clzll((long long)(keys[i] ^ keys[j]));
}

constexpr uint32_t REFIT_BLOCK = 256;     // leaves per block of k_bvh_refit

// (also lists the inner nodes whose leaves reach across blocks of k_bvh_refit: top_list[0 .. *top_count), any order)
__global__ void k_bvh_split(uint32_t n, const unsigned long long *__restrict__ keys, uint32_t *__restrict__ child,
                            uint32_t *__restrict__ parent, uint32_t *__restrict__ range, uint32_t *__restrict__ top_list,
                            uint32_t *__restrict__ top_count)
{
    int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    int N = (int)n;
    if (i >= N - 1) return;
    int d = (bvh_delta(keys, N, i, i + 1) - bvh_delta(keys, N, i, i - 1)) >= 0 ? 1 : -1;
    int dmin = bvh_delta(keys, N, i, i - d);
    int lmax = 2;
    while (bvh_delta(keys, N, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (bvh_delta(keys, N, i, i + (l + t) * d) > dmin) l += t;
    int j = i + l * d;
    int dnode = bvh_delta(keys, N, i, j);
    int s = 0, t = l;
    do {
        t = (t + 1) / 2;
        if (bvh_delta(keys, N, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    int gamma = i + s * d + min(d, 0);
    int first = min(i, j), last = max(i, j);
    uint32_t left = (first == gamma) ? (uint32_t)(N - 1 + gamma) : (uint32_t)gamma;
    uint32_t right = (last == gamma + 1) ? (uint32_t)(N - 1 + gamma + 1) : (uint32_t)(gamma + 1);
    child[2 * i] = left; child[2 * i + 1] = right;
    parent[left] = (uint32_t)i; parent[right] = (uint32_t)i;
    range[2 * i] = (uint32_t)first; range[2 * i + 1] = (uint32_t)last;
    if (i == 0) parent[0] = 0xffffffffu;
    if ((uint32_t)first / REFIT_BLOCK != (uint32_t)last / REFIT_BLOCK) top_list[atomicAdd(top_count, 1u)] = (uint32_t)i;
}

// Boxes (and normal boxes) bottom-up in two launches.  k_bvh_refit: a block owns REFIT_BLOCK consecutive leaves and computes
// every inner node whose leaves all belong to it; the second thread to arrive at a node computes it, and since both came
// from this block a workgroup-scope fence orders the hand-over.  Nodes that reach across blocks (about two per block)
// are left to k_bvh_refit_top, ONE block that starts after the first launch has finished (so everything below is
// visible) and relaxes them round by round with barriers.  (One launch with device-scope fences at those nodes took
// 0.34-0.66 ms for 100 k triangles: on this chip a device-scope release writes back an XCD's L2, and the path to the root is a
// chain of them.)
DEV void bvh_refit_node(uint32_t cur, const uint32_t *child, float4 *box_lo, float4 *box_hi, float4 *nrm_lo, float4 *nrm_hi) {
    uint32_t a = child[2 * cur], b = child[2 * cur + 1];
    float4 la = box_lo[a], ha = box_hi[a], lb = box_lo[b], hb = box_hi[b];
    box_lo[cur] = make_float4(fminf(la.x, lb.x), fminf(la.y, lb.y), fminf(la.z, lb.z), fmaxf(la.w, lb.w));
    box_hi[cur] = make_float4(fmaxf(ha.x, hb.x), fmaxf(ha.y, hb.y), fmaxf(ha.z, hb.z), 0.f);
    float4 na = nrm_lo[a], ma = nrm_hi[a], nb = nrm_lo[b], mb = nrm_hi[b];
    nrm_lo[cur] = make_float4(fminf(na.x, nb.x), fminf(na.y, nb.y), fminf(na.z, nb.z), fminf(na.w, nb.w));
    nrm_hi[cur] = make_float4(fmaxf(ma.x, mb.x), fmaxf(ma.y, mb.y), fmaxf(ma.z, mb.z), 0.f);
}

// inner node `i` has all its leaves in one block of k_bvh_refit
DEV bool bvh_refit_local(uint32_t i, const uint32_t *range) { return range[2 * i] / REFIT_BLOCK == range[2 * i + 1] / REFIT_BLOCK; }

__global__ __launch_bounds__(REFIT_BLOCK) void k_bvh_refit(uint32_t n, const uint32_t *__restrict__ child, const uint32_t *__restrict__ parent,
                            const uint32_t *__restrict__ range, uint32_t *__restrict__ arrived, float4 *box_lo, float4 *box_hi,
                            float4 *nrm_lo, float4 *nrm_hi)
{
    uint32_t k = blockIdx.x * REFIT_BLOCK + threadIdx.x;
    if (k >= n || n < 2) return;
    uint32_t cur = parent[(n - 1) + k];
    while (cur != 0xffffffffu && bvh_refit_local(cur, range)) {
        __threadfence_block();
        if (atomicAdd(&arrived[cur], 1u) == 0u) return;          // the sibling subtree is not finished yet
        __threadfence_block();
        bvh_refit_node(cur, child, box_lo, box_hi, nrm_lo, nrm_hi);
        cur = parent[cur];
    }
}

// `done` = the array `arrived` of the first launch, reused: a node of this launch is done when its entry is 3.  One block;
// a round computes every listed node whose children are ready, rounds until none is left (as many as the listed nodes are deep).
__global__ __launch_bounds__(1024) void k_bvh_refit_top(uint32_t n, const uint32_t *__restrict__ child, const uint32_t *__restrict__ range,
                                                          const uint32_t *__restrict__ top_list, const uint32_t *__restrict__ top_count,
                                                          uint32_t *done, float4 *box_lo, float4 *box_hi, float4 *nrm_lo, float4 *nrm_hi)
{
    __shared__ uint32_t pending;
    if (n < 2) return;
    const uint32_t n_inner = n - 1u, count = *top_count;
    // (every round completes at least one listed node of a well-formed tree; the bound ends the loop for any other input)
    for (uint32_t round = 0; round <= count; round++) {
        if (threadIdx.x == 0) pending = 0u;
        __syncthreads();
        for (uint32_t j = threadIdx.x; j < count; j += blockDim.x) {
            const uint32_t i = top_list[j];
            if (done[i] == 3u) continue;
            const uint32_t a = child[2 * i], b = child[2 * i + 1];
            const bool ra = a >= n_inner || bvh_refit_local(a, range) || done[a] == 3u;
            const bool rb = b >= n_inner || bvh_refit_local(b, range) || done[b] == 3u;
            if (ra && rb) {
                __threadfence_block();           // the children's boxes are read after their flags
                bvh_refit_node(i, child, box_lo, box_hi, nrm_lo, nrm_hi);
                __threadfence_block();           // and this node's flag is set after its boxes
                done[i] = 3u;
            }
            else pending = 1u;
        }
        __threadfence_block();
        __syncthreads();
        if (pending == 0u) break;
        __syncthreads();
    }
}

DEV uint32_t bvh_ref_of(uint32_t node, uint32_t n, const uint32_t *range) {
    if (node >= n - 1) return BVH_LEAF | (1u << 28) | (node - (n - 1));
    uint32_t first = range[2 * node], cnt = range[2 * node + 1] - first + 1;
    if (cnt <= (uint32_t)BVH_LEAF_MAX) return BVH_LEAF | (cnt << 28) | first;
    return node;
}

// Cone of the unit normals inside the box [nl, nh], scaled for bvh_box: returns {a * k, sin * k}
// with k = cos_bound * sine_min / BVH_TIGHT, or a record that disables the tight bound.
DEV float4 bvh_cone(float4 nl, float4 nh) {
    f3 cen = mk(0.5f * (nl.x + nh.x), 0.5f * (nl.y + nh.y), 0.5f * (nl.z + nh.z));
    f3 half = mk(0.5f * (nh.x - nl.x), 0.5f * (nh.y - nl.y), 0.5f * (nh.z - nl.z));
    float len = sqrtf(dot(cen, cen));
    float spread = sqrtf(dot(half, half)) * 1.0001f + 1.0e-6f;
    // every unit normal u in the box has dot(u, cen/len) >= len - spread
    float ca = fminf(len - spread, 1.0f), smin = nl.w;
    if (!(ca > 1.0e-3f) || !(smin > 1.0e-6f)) return make_float4(0.f, 0.f, 0.f, 3.0e38f);
    float sa = sqrtf(fmaxf(0.f, 1.0f - ca * ca)) * 1.0001f + 1.0e-7f;
    float k = smin / BVH_TIGHT;
    f3 a = cen * (1.0f / len);
    return make_float4(a.x * (ca * k), a.y * (ca * k), a.z * (ca * k), sa * k);
}

// Upper 16 bits of the smallest bfloat16 that is >= x, for finite x >= 0 (the largest finite bfloat16 beyond that).
DEV uint32_t bf16_up(float x) {
    uint32_t u = (__float_as_uint(x) + 0xffffu) >> 16;
    return u > 0x7f7fu ? 0x7f7fu : u;
}

// next float above a half extent h >= 0 (covers the rounding of the subtraction that made it); 0 stays 0
DEV float half_up(float h) { return h > 0.0f ? __uint_as_float(__float_as_uint(h) + 1u) : 0.0f; }

// A node lists up to four descendants (its children, the larger ones opened once more) and carries one
// cone and one A, those of all its triangles.  (A cone per slot prunes a little more but the larger record costs
// more than that saves; a ninth float4 for the cone, with A per slot, made every node visit touch two cache lines.)
__global__ void k_bvh_emit(uint32_t n, const uint32_t *__restrict__ child, const uint32_t *__restrict__ range,
                            const float4 *__restrict__ box_lo, const float4 *__restrict__ box_hi,
                            const float4 *__restrict__ nrm_lo, const float4 *__restrict__ nrm_hi, float4 *__restrict__ nodes)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (n < 2 || i >= n - 1) return;
    // start from the two children and twice replace the inner slot with the largest box by its two children
    uint32_t slot[4] = {child[2 * i], child[2 * i + 1], 0u, 0u};
    int ns = 2;
    for (int round = 0; round < 2; round++) {
        int pick = -1; float best_area = -1.0f; bool best_mixed = false;
        for (int k = 0; k < ns; k++) {
            if (bvh_ref_of(slot[k], n, range) & BVH_LEAF) continue;
            float4 lo = box_lo[slot[k]], hi = box_hi[slot[k]];
            float dx = hi.x - lo.x, dy = hi.y - lo.y, dz = hi.z - lo.z;
            float area = dx * dy + dy * dz + dz * dx;
            // a child without a normal cone (it mixes orientation classes) is opened before any other: as a node of its own
            // it would take the loose enlargement for all its slots, and every ray that reaches it would enter them all
            const bool mixed = bvh_cone(nrm_lo[slot[k]], nrm_hi[slot[k]]).w >= 3.0e38f;
            if ((mixed && !best_mixed) || (mixed == best_mixed && area > best_area)) { best_area = area; pick = k; best_mixed = mixed; }
        }
        if (pick < 0) break;
        uint32_t c = slot[pick];
        slot[pick] = child[2 * c];
        slot[ns++] = child[2 * c + 1];
    }
    // Slots still free (the node's subtree has only leaves left): a walk tests all four slots of a record anyway, so a
    // free slot costs what a used one does.  Opening two-triangle leaves into their two triangles, largest box first,
    // while slots are free gives each triangle a box of its own.  Measured (off by default): 15-30 % fewer exact tests
    // per ray and frames 0.5-1.5 % SLOWER: a wave pays per step, and two one-triangle leaves are two leaf steps where
    // the two-triangle leaf was one.
    if (BVH_SPLIT_LEAVES) {
        while (ns < 4) {
            int pick = -1; float best_area = -1.0f;
            for (int k = 0; k < ns; k++) {
                if (slot[k] >= n - 1) continue;                           // a single triangle already
                float4 lo = box_lo[slot[k]], hi = box_hi[slot[k]];
                float dx = hi.x - lo.x, dy = hi.y - lo.y, dz = hi.z - lo.z;
                float area = dx * dy + dy * dz + dz * dx;
                if (area > best_area) { best_area = area; pick = k; }
            }
            if (pick < 0) break;
            uint32_t c = slot[pick];
            slot[pick] = child[2 * c];
            slot[ns++] = child[2 * c + 1];
        }
    }
    // what belongs to the node as a whole travels in the w words of the four lo entries, so that the record is one
    // 128-byte cache line: A of all the node's triangles (the largest of its slots' values) and the cone
    const float4 cone = bvh_cone(nrm_lo[i], nrm_hi[i]);
    const float w[4] = {__uint_as_float((bf16_up(box_lo[i].w) << 16) | bf16_up(cone.w)), cone.x, cone.y, cone.z};
    float4 *N = nodes + BVH_NODE_F4 * (size_t)i;
    for (int k = 0; k < 4; k++) {
        float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = make_float4(0.f, 0.f, 0.f, __uint_as_float(BVH_NONE));   // unused slot
        if (k < ns) {
            // centre and half extent, the half extent rounded up so that centre +- half extent contains the box
            const float4 bl = box_lo[slot[k]], bh = box_hi[slot[k]];
            lo = make_float4(0.5f * (bl.x + bh.x), 0.5f * (bl.y + bh.y), 0.5f * (bl.z + bh.z), 0.f);
            hi = make_float4(half_up(fmaxf(bh.x - lo.x, lo.x - bl.x)), half_up(fmaxf(bh.y - lo.y, lo.y - bl.y)),
                             half_up(fmaxf(bh.z - lo.z, lo.z - bl.z)), __uint_as_float(bvh_ref_of(slot[k], n, range)));
        }
        lo.w = w[k];
        N[k] = lo; N[4 + k] = hi;
    }
}

// ---------------------------------------------------------------------------------------
// traversal
// ---------------------------------------------------------------------------------------
// Per-ray constants of the box test.  A box is enlarged on every side by `infl`, chosen so that
// the point o + t*d of every hit an EXACT test can accept for a triangle inside the box lies
// inside the enlarged box.  With q = o - p, s = sine of the angle between e1 and e2 and
// c = |cos| of the angle between d and the triangle normal, |det| = |d||e1||e2| s c, and the
// rounding analysis of the tests gives (k <= 8 operations per determinant, eps = 2^-24):
//   numerators  : |delta u||e1| + |delta v||e2| <= 2 * 8 eps |d||e1||e2||q| / |det|
//   denominator : relative error of det <= 4 eps |d||e1||e2| s / |det|, moving the point by at
//                 most (|e1| + |e2|) times that
//   ray parameter t: the same two forms, times |d|
// In a box whose corners are at most qb (L1) from o, |q| <= qb and |e1|, |e2| <= 2 qb, so
//   error <= 1.92e-6 * qb / (s c)                         ("tight", scale free)
//   error <= 1.92e-2 * |d| |e1||e2| * qb                  ("loose", from |det| > 1e-4)
// (qb per box rather than one bound per ray: shadow rays start on the surface, where the loose
// form would otherwise enlarge the boxes next to the origin by the size of the whole scene)
// Both are used with margin: A = BVH_LOOSE * max |e1||e2| over the subtree, and the tight form
// with BVH_TIGHT when the normals of the subtree lie in a cone that keeps s*c >= 2e-3 for this
// ray (g below is a lower bound of s*c / BVH_TIGHT).  E covers the rounding of the slab
// arithmetic itself (each face moves by less than 2^-22 of |o| + |box coordinate|).  Boxes
// are stored as centre c and half extent h >= 0, and the two faces of an axis are evaluated
// as fma(-+(h + infl), |inv|, (c - o) * inv) (bvh_box).  What keeps that conservative when a
// direction component is (nearly) zero: an infinite inv makes every term of the axis +-inf or
// NaN, and a NaN drops out of the max3 / min3, so the axis merely stops culling; a huge
// finite inv would let (c - o) * inv overflow while the half-extent term stays finite (an
// empty interval for a box the ray is inside of), so tray_make clamps |inv| to inv_clamp, a
// value that every product with |c - o| <= Q survives and that still reaches past every
// parameter at which a hit can be accepted (inv_clamp * |d| >= 6e6, see build_bvh).
struct TRay { f3 o, inv, dhat; float dlen, E; };

DEV TRay tray_make(const DScene &S, const DBvh &B, f3 o, f3 d) {
    TRay r;
    r.o = o;
    // A direction component of (nearly) zero makes its reciprocal huge or infinite.  An infinite one is harmless: every
    // term of that axis is then +-inf or NaN, and a NaN drops out of the max3 / min3 of bvh_box (the axis no longer culls).
    // A huge finite one is not: centre * reciprocal overflows to inf while the fused half-extent term stays finite, and
    // the axis would wrongly report an empty interval.  So reciprocals are limited to +-inv_clamp where products with them
    // stay finite (|centre - origin| <= Q, the ray's distance scale), and become infinite where they would not:
    //  * clamped, an origin outside such a slab still gives an empty interval (of the order of inv_clamp * distance, far
    //    beyond any ray parameter), an origin inside one that reaches past every parameter at which a hit can be accepted;
    //    for that the clamp must exceed 6e6 / |d| for the shortest direction that can still pass the |det| thresholds,
    //    |d| >= 1e-4 / (|e1||e2|): build_bvh sets inv_clamp from the scene's size;
    //  * where inv_clamp * Q leaves the float range (astronomically large scenes, far-away origins) every reciprocal above
    //    1e37 / Q counts as infinite, and axis-parallel rays lose that axis' culling.
    // (hardware reciprocal and square root, one unit in the last place: nothing here decides a hit, and E leaves room
    // for it; the correctly rounded forms cost ten instructions each)
    const float Q = ((fabsf(o.x) + fabsf(o.y)) + fabsf(o.z)) + B.scene_mag;
    const float lim = 1.0e37f * __builtin_amdgcn_rcpf(Q), ic = B.inv_clamp;
    const float thr = fminf(ic, lim), cap = (ic <= lim) ? ic : __builtin_inff();
    float vx = __builtin_amdgcn_rcpf(d.x), vy = __builtin_amdgcn_rcpf(d.y), vz = __builtin_amdgcn_rcpf(d.z);
    r.inv = mk(fabsf(vx) > thr ? copysignf(cap, vx) : vx, fabsf(vy) > thr ? copysignf(cap, vy) : vy,
               fabsf(vz) > thr ? copysignf(cap, vz) : vz);
    r.dlen = __builtin_amdgcn_sqrtf(dot(d, d));
    float il = __builtin_amdgcn_rcpf(r.dlen);
    r.dhat = d * il;
    if (!(r.dlen < 3.0e38f) || !(il < 3.0e38f)) { r.dlen = 3.0e38f; r.dhat = mk(0.f, 0.f, 0.f); }   // loose bound only
    r.E = 1.0e-6f * Q + 1.0e-30f;
    return r;
}

// Growth of a node's boxes per unit of L1 distance, from the node's w words: the smaller of the loose form A * |d| and,
// where the node's cone keeps the ray away from the planes of its triangles (g >= BVH_TIGHT_MIN), the tight form 1 / g.
DEV float bvh_node_growth(const TRay &r, float w0f, float kx, float ky, float kz) {
    const uint32_t w0 = __float_as_uint(w0f);
    const float ad = __uint_as_float(w0 & 0xffff0000u) * r.dlen;
    const float g = fabsf(fmaf(kz, r.dhat.z, fmaf(ky, r.dhat.y, kx * r.dhat.x))) - __uint_as_float(w0 << 16);
    return (g >= BVH_TIGHT_MIN) ? fminf(ad, __builtin_amdgcn_rcpf(g)) : ad;
}

// entry parameter of the enlarged box if the half-line t >= 0 may reach it at t <= limit; the box is centre `cw` +-
// half extent `hr`; `m` = min(A * |d|, 1/g) of the node the box belongs to (1/g = +inf where the tight form does not
// apply), so that the box grows by m times the L1 distance of its farthest corner, plus E.  With the box as centre and
// half extent the two faces of an axis are tc -+ th, th >= 0, already in order: no min/max per axis.  A NaN (inf - inf
// or 0 * inf, unclamped reciprocals only) drops out of the min3 / max3, which can only widen the interval.
template <bool BOUNDED = true>
DEV bool bvh_box(const TRay &r, float4 cw, float4 hr, float m, float limit, float &tnear) {
    float cx = cw.x - r.o.x, cy = cw.y - r.o.y, cz = cw.z - r.o.z;
    float qb = ((fabsf(cx) + hr.x) + (fabsf(cy) + hr.y)) + (fabsf(cz) + hr.z);     // L1 distance of the farthest corner
    float in = fmaf(qb, m, r.E);
    float hx = hr.x + in, hy = hr.y + in, hz = hr.z + in;
    float tcx = cx * r.inv.x, tcy = cy * r.inv.y, tcz = cz * r.inv.z;
    float ax = fabsf(r.inv.x), ay = fabsf(r.inv.y), az = fabsf(r.inv.z);
    float tmin = max3f(fmaf(-hx, ax, tcx), fmaf(-hy, ay, tcy), fmaf(-hz, az, tcz));
    float tmax = min3f(fmaf(hx, ax, tcx), fmaf(hy, ay, tcy), fmaf(hz, az, tcz));
    tnear = tmin;
#if RT_BOX_ONE_CMP
    // One comparison instead of three, each of which is a lane mask in scalar registers that a scalar AND has to wait
    // for: max(tmin, 0) <= min(tmax, limit) is the same decision for 0 <= limit (BOUNDED false: the limit is +inf), and
    // where a NaN makes it differ the box is entered, never skipped.  The same count of vector instructions, two scalar
    // ones and their waits fewer per slot: camera kernel -2.7 %, C2 -0.9 %, C5 -1.1 % (a wave alone issues the
    // three-comparison form at 41 cycles per slot decision, this one at 17).
    return fmaxf(tmin, 0.0f) <= (BOUNDED ? fminf(tmax, limit) : tmax);
#else
    return (tmin <= tmax) && (tmax >= 0.0f) && !(tmin > limit);
#endif
}

// EXACT tests on a {p, e1, e2, c21} record: the bits of tri_exact_cpu / tri_exact_cl with
// cross(e2, e1) read from the record (the same operations on the same inputs) and
// cross(e1, e2) taken as its negation (exact up to the sign of zeros, which cannot change
// an accepted hit: a zero dett is rejected by both t thresholds).
// Most candidates are misses, so each quotient is first bounded without dividing: with
// a = |det| and the numerators multiplied by sign(det),
//   dett <= T/2 * a          =>  fl(dett/det) <= T        (T = 1e-4 / 1e-3, rejected by t > T)
//   detu <  -1e-30 * a       =>  fl(detu/det) < 0         (likewise detv)
//   detu + detv > 1.00001 a  =>  fl(fl(u) + fl(v)) > 1
// (margins far above the 2^-24 roundings, and a > 1e-4 keeps every product normal).  What is
// not rejected this way is decided by the reference's own divisions and comparisons.
template <int MODE>
DEV bool tri_exact_rec(f3 o, f3 d, f3 p, f3 e1, f3 e2, f3 c21, float &t, float &u, float &v) {
    f3 q = o - p;
    float det = dot(d, c21);
    float a = fabsf(det);
    if (MODE == HIPEXEC_MODE_CL) { if (!(a > 0.001f)) return false; }
    else if (a <= RT_EPSF) return false;
    const bool neg = det < 0.0f;
    float dett = dot(q, -c21);
    if ((neg ? -dett : dett) <= (MODE == HIPEXEC_MODE_CL ? 0.0005f : 0.00005f) * a) return false;
    float detu = dot(d, cross(e2, q));
    if ((neg ? -detu : detu) < -1.0e-30f * a) return false;
    float detv = dot(d, cross(q, e1));
    if ((neg ? -detv : detv) < -1.0e-30f * a) return false;
    if ((neg ? -(detu + detv) : (detu + detv)) > 1.00001f * a) return false;
    float tt = dett / det, uu = detu / det, vv = detv / det;
    if (MODE == HIPEXEC_MODE_CL) {
        bool hit = (det > 0.001f || det < -0.001f);
        hit = hit && (tt > 0.001f);
        hit = hit && (uu > 0.0f && vv > 0.0f && uu + vv < 1.0f);
        t = tt; u = uu; v = vv;
        return hit;
    }
    if (tt <= RT_EPSF) return false;
    if (uu < 0.0f || vv < 0.0f || vv + uu > 1.0f) return false;
    t = tt; u = uu; v = vv;
    return true;
}

// winner rule of the linear scans made explicit: smaller t, then smaller triangle index; the CPU tracer accepts a
// candidate only with t > 0 (ray_tracer.cpp:214), which also keeps out a NaN t (overflowing determinants pass every
// threshold comparison of the exact test), the OpenCL executor takes every flagged hit (opencl_executor.cpp:242)
template <int MODE>
DEV void best_update_indexed(Best &b, float t, float u, float v, int id) {
    if (MODE != HIPEXEC_MODE_CL && !(t > 0)) return;
    if (b.id < 0 || t < b.t || (t == b.t && id < b.id)) { b.t = t; b.u = u; b.v = v; b.id = id; }
}

// Type of the per-lane flags of the traversal loops (found, overflow): 0/1 words, see the note above TState.
#ifdef RT_FLAG_BOOL
typedef bool rt_flag_t;              // forensic build only (tools/flag_forensics.py): the form that lost updates
#else
typedef uint32_t rt_flag_t;
#endif

// One inner-node step of a lane: the four slot boxes of record N against the ray, the nearest hit slot
// becomes the lane's node, the others go on its stack (farthest first).
template <bool UNIFORM, bool SORT = true, int BS = 256>
DEV void bvh_inner_step(const DBvh &B, const TRay &r, const float4 *__restrict__ N, float limit, uint32_t *stk, uint32_t &node, int &sp,
                        rt_flag_t &overflow)
{
    if (UNIFORM) asm volatile("; wave-uniform record" ::: );          // keeps the two instances apart (scalar vs vector fetch)
    float4 lo[4], hi[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { lo[k] = N[k]; hi[k] = N[4 + k]; }
    const float m = bvh_node_growth(r, lo[0].w, lo[1].w, lo[2].w, lo[3].w);
    float tk[4]; bool hk[4];
#pragma unroll
    // (unused slots marked by half extents of -inf, which fail the box test by themselves, instead of this look at the
    // reference: C2 and C3 2.5 % slower, measured twice, a round apart)
    for (int k = 0; k < 4; k++) hk[k] = bvh_box<SORT>(r, lo[k], hi[k], m, limit, tk[k]) && __float_as_uint(hi[k].w) != BVH_NONE;
    if (!SORT) {
        // Any-hit rays: the answer does not depend on the order of the walk, and ordering the slots by entry distance
        // (five compare-exchanges) costs more than it finds occluders sooner (measured: C5 -8 %, C2 -2.5 % without it).
        // The first hit slot is entered next, the others go on the stack.
        uint32_t nxt = BVH_NONE;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (hk[k]) {
                const uint32_t ref = __float_as_uint(hi[k].w);
                if (nxt == BVH_NONE) nxt = ref;
                else if (sp < B.stack_limit) { stk[BS * sp] = ref; sp++; }
                else overflow = 1u;
            }
        }
        if (overflow) node = BVH_NONE;
        else if (nxt != BVH_NONE) node = nxt;
        else if (sp > 0) { sp--; node = stk[BS * sp]; }
        else node = BVH_NONE;
        return;
    }
    // hit slots in order of entry: the nearest is visited next, the others go on the stack, farthest first
    float key[4]; uint32_t ref[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { key[k] = hk[k] ? fminf(tk[k], 3.0e38f) : __builtin_inff(); ref[k] = hk[k] ? __float_as_uint(hi[k].w) : BVH_NONE; }
#define BVH_CSWAP(a, b) { bool sw = key[b] < key[a]; float ka = sw ? key[b] : key[a], kb = sw ? key[a] : key[b]; \
              uint32_t ra = sw ? ref[b] : ref[a], rb = sw ? ref[a] : ref[b]; key[a] = ka; key[b] = kb; ref[a] = ra; ref[b] = rb; }
    BVH_CSWAP(0, 1) BVH_CSWAP(2, 3) BVH_CSWAP(0, 2) BVH_CSWAP(1, 3) BVH_CSWAP(1, 2)
#undef BVH_CSWAP
#pragma unroll
    for (int k = 3; k >= 1; k--) {
        if (ref[k] != BVH_NONE) {
            if (sp < B.stack_limit) { stk[BS * sp] = ref[k]; sp++; }
            else overflow = 1u;
        }
    }
    if (overflow) node = BVH_NONE;
    else if (ref[0] != BVH_NONE) node = ref[0];
    else if (sp > 0) { sp--; node = stk[BS * sp]; }
    else node = BVH_NONE;
}

// ---------------------------------------------------------------------------------------
// Burst form of the traversal for kernels that refill idle lanes: the lanes of a wave whose
// `node` is not BVH_NONE advance until fewer than `min_active` of them are left, so that the
// caller can hand new rays to the idle lanes before the wave runs mostly empty.
// ---------------------------------------------------------------------------------------
// The per-lane flags are 0/1 words on purpose.  As `bool` the compiler keeps `found` as a lane mask in scalar
// registers, and in an earlier shadow kernel whose lanes refilled from a block list (this burst loop inlined into a
// refill loop) the -O3 build lost occluders: reproduced with `found` alone as bool, gone at -O1 and with word flags,
// unchanged when SGPR spills go to memory instead of VGPR lanes.  `found` is lane-local and every cross-lane operation
// of those loops runs with the wave converged, so the source has no undefined behaviour for an optimiser to use: a
// code-generation defect of ROCm 7.2's hipcc for an i1 value live across nested divergent loops.  Today's kernels
// give identical results in both forms (-DRT_FLAG_BOOL builds the other one for the regression test).
struct TState { uint32_t node; int sp; float limit; rt_flag_t found, overflow; };
DEV void bvh_count(unsigned long long *counters, int klass, uint32_t n_nodes, uint32_t n_tris, uint32_t n_rays);

// Every iteration the wave takes ONE of two steps, whichever more of its lanes are waiting for:
// the inner-node step (four box tests, push/pop) or the leaf step (exact tests).  Lanes waiting for
// the other kind sit the iteration out.  (`sched` = 0 selects the classic while-while order instead:
// inner nodes until no lane has one, then leaves.)
template <int MODE, bool ANY, int BS = 256>
DEV void bvh_burst(const DBvh &B, const TRay &r, f3 o, f3 d, TState &st, Best &best, uint32_t *stk, int min_active,
                   uint32_t &n_nodes, uint32_t &n_tris, int sched = 8, const float4 *__restrict__ nodes_sc = nullptr)
{
    const float4 *__restrict__ T = B.tris;
    uint32_t node = st.node;
    int sp = st.sp;
    float limit = st.limit;
    rt_flag_t found = st.found, overflow = st.overflow;
    for (;;) {
        const bool at_leaf = (node & BVH_LEAF) != 0u;                    // BVH_NONE has the bit clear
        const bool at_inner = !at_leaf && node != BVH_NONE;
        const int n_inner = __popcll(__ballot(at_inner)), n_leaf = __popcll(__ballot(at_leaf));
        if (n_inner + n_leaf == 0 || n_inner + n_leaf < min_active) break;
        const bool inner_step = sched ? (n_inner * 8 >= n_leaf * sched) : (n_inner > 0);
        if (inner_step) {
            // Neighbouring rays mostly sit on the same node near the top of the tree: if every lane of this step does, the
            // record comes through the scalar cache (one fetch per wave) and the texture-address units are left to the
            // steps whose lanes have spread out.
            const unsigned long long m_inner = __ballot(at_inner);
            // (v_readlane with the scalar lane number: a shuffle here is a ds_bpermute, a trip through the LDS pipe in every step)
            const uint32_t lead_node = (uint32_t)__builtin_amdgcn_readlane((int)node, __ffsll((long long)m_inner) - 1);
            // (nodes_sc: the node array as a restrict-qualified kernel argument; only then can the compiler prove that
            // the kernel's own stores do not touch it and use scalar loads)
            const bool uniform = nodes_sc != nullptr && __ballot(at_inner && node == lead_node) == m_inner;
            if (uniform) {
                const uint32_t un = (uint32_t)__builtin_amdgcn_readfirstlane((int)lead_node);
                if (at_inner) { n_nodes++; bvh_inner_step<true, !ANY, BS>(B, r, nodes_sc + BVH_NODE_F4 * (size_t)un, limit, stk, node, sp, overflow); }
            } else if (at_inner) {
                n_nodes++;
                bvh_inner_step<false, !ANY, BS>(B, r, B.nodes + BVH_NODE_F4 * (size_t)node, limit, stk, node, sp, overflow);
            }
        } else if (at_leaf) {
            uint32_t first = node & 0x0fffffffu, cnt = (node >> 28) & 7u;
            // the records of the first two triangles are requested together (the array is padded by one record), so a
            // two-triangle leaf waits for memory once
            float4 q[6];
            // The lane's next node, unless this leaf ends its ray, is the top of its stack: read now, the LDS round trip
            // passes while the triangle records arrive instead of after the tests (C2 -0.8 %, C3 -1.1 %, C5 -2.1 %; the
            // same early read in every inner-node step, or in the leaf step of the packets, whose scalar loads share
            // the wait with it, is slower).
            const uint32_t below = sp > 0 ? stk[BS * (sp - 1)] : BVH_NONE;
            // (closest-hit rays ask for the second record only where the leaf has one: leaves of a single triangle are common
            // and that kernel keeps the texture addressers busiest, bounce class -3.4 %; shadow rays ask for both at once
            // whatever the count: conditional, C5 is 0.6 % slower)
#pragma unroll
            for (int i = 0; i < 3; i++) q[i] = T[3 * (size_t)first + i];
#pragma unroll
            for (int i = 3; i < 6; i++) q[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ANY || cnt >= 2u) {
#pragma unroll
                for (int i = 3; i < 6; i++) q[i] = T[3 * (size_t)first + i];
            }
            for (uint32_t k = first; k < first + cnt; k++) {
                if (ANY && found) continue;
                float4 e0, e1, e2;
                if (k == first) { e0 = q[0]; e1 = q[1]; e2 = q[2]; }
                else if (k == first + 1u) { e0 = q[3]; e1 = q[4]; e2 = q[5]; }
                else { e0 = T[3 * (size_t)k]; e1 = T[3 * (size_t)k + 1]; e2 = T[3 * (size_t)k + 2]; }
                n_tris++;
                float t, u, v;
                if (tri_exact_rec<MODE>(o, d, xyz(e0), xyz(e1), xyz(e2), mk(e0.w, e1.w, e2.w), t, u, v)) {
                    if (ANY) found = 1u;
                    else {
                        best_update_indexed<MODE>(best, t, u, v, (int)B.ids[k]);
                        limit = fminf(limit, best.t);
                    }
                }
            }
            if (ANY && found) node = BVH_NONE;
            else { node = below; sp = sp > 0 ? sp - 1 : 0; }
        }
    }
    st.node = node; st.sp = sp; st.limit = limit; st.found = found; st.overflow = overflow;
}

// linear scan of all records: the answer for a ray whose traversal stack overflowed
template <int MODE, bool ANY>
DEV bool bvh_linear(const DBvh &B, f3 o, f3 d, Best &best, uint32_t &n_tris) {
    const float4 *__restrict__ T = B.tris;
    bool found = false;
    if (!ANY) best = best_none();
    for (uint32_t k = 0; k < B.n && !(ANY && found); k++) {
        float4 e0 = T[3 * (size_t)k], e1 = T[3 * (size_t)k + 1], e2 = T[3 * (size_t)k + 2];
        n_tris++;
        float t, u, v;
        if (tri_exact_rec<MODE>(o, d, xyz(e0), xyz(e1), xyz(e2), mk(e0.w, e1.w, e2.w), t, u, v)) {
            if (ANY) found = true;
            else best_update_indexed<MODE>(best, t, u, v, (int)B.ids[k]);
        }
    }
    return found;
}

// ---------------------------------------------------------------------------------------
// Packet form of the traversal for the 64 rays of a wave that travel together (the camera rays of an
// 8x8-pixel tile, shadow rays from neighbouring surface points to one lamp).  The wave walks ONE path
// through the hierarchy: node and leaf records are fetched once per wave through the scalar cache, every
// lane tests them against its own ray, and the walk enters a slot if any lane's ray may.  Each stack
// entry carries the mask of the lanes that asked for it, so a lane evaluates exactly the boxes and exact
// tests its own walk would have (in a different order, which the winner rule does not depend on).
// The stack (node, 64-bit lane mask) is per wave, in LDS.  Returns false if that stack overflowed; the caller then
// repeats the rays with the per-lane traversal (closest: `best` may hold valid candidates and stays usable).
// ---------------------------------------------------------------------------------------
constexpr int PK_STACK = 4 * BVH_STACK;   // entries of the wave's stack: up to three pushes per level

template <int MODE, bool ANY>
DEV bool bvh_packet(const DBvh &B, const TRay &r, f3 o, f3 d, bool valid, float &limit, Best &best, uint32_t &found,
                    uint32_t *pst, uint32_t &n_nodes, uint32_t &n_tris)
{
    const float4 *__restrict__ T = B.tris;
    const float4 *__restrict__ NB = B.nodes;
    const int lane = lane_id();
    unsigned long long alive = __ballot(valid && !(ANY && found));
    uint32_t node = B.root;
    unsigned long long mask = alive;
    int sp = 0;
    bool ok = true;
    const int pk_limit = B.stack_limit * (PK_STACK / BVH_STACK);     // PK_STACK unless the test hook shortens the stacks
    for (;;) {
        // node and mask are the same in every lane; saying so lets the records below come through the scalar cache
        node = (uint32_t)__builtin_amdgcn_readfirstlane((int)node);
        mask = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(mask >> 32)) << 32) |
               (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)mask);
        mask &= alive;
        if (mask == 0ull || node == BVH_NONE) {
            if (sp == 0) break;
            sp--;
            node = (uint32_t)__builtin_amdgcn_readfirstlane((int)pst[sp]);
            uint32_t mlo = (uint32_t)__builtin_amdgcn_readfirstlane((int)pst[PK_STACK + sp]);
            uint32_t mhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)pst[2 * PK_STACK + sp]);
            mask = ((unsigned long long)mhi << 32) | mlo;
            continue;
        }
        const bool mine = ((mask >> lane) & 1ull) != 0ull;
        if (node & BVH_LEAF) {
            const uint32_t first = node & 0x0fffffffu, cnt = (node >> 28) & 7u;
            for (uint32_t k = first; k < first + cnt; k++) {
                float4 e0 = T[3 * (size_t)k], e1 = T[3 * (size_t)k + 1], e2 = T[3 * (size_t)k + 2];
                if (mine && !(ANY && found)) {
                    n_tris++;
                    float t, u, v;
                    if (tri_exact_rec<MODE>(o, d, xyz(e0), xyz(e1), xyz(e2), mk(e0.w, e1.w, e2.w), t, u, v)) {
                        if (ANY) found = 1u;
                        else {
                            best_update_indexed<MODE>(best, t, u, v, (int)B.ids[k]);
                            limit = fminf(limit, best.t);
                        }
                    }
                }
            }
            if (ANY) alive &= ~__ballot(found != 0u);
            node = BVH_NONE;
        } else {
            const float4 *N = NB + BVH_NODE_F4 * (size_t)node;
            float4 lo[4], hi[4];
#pragma unroll
            for (int k = 0; k < 4; k++) { lo[k] = N[k]; hi[k] = N[4 + k]; }
            float tk[4]; bool hk[4] = {false, false, false, false};
            if (mine) {
                n_nodes++;
                const float m = bvh_node_growth(r, lo[0].w, lo[1].w, lo[2].w, lo[3].w);
                // The record sits in scalar registers, so an unused slot (29 % of them with two triangles per leaf) is
                // skipped by a scalar branch.  (The same branch in the uniform steps of the shadow rays: C5 +2 %.)
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    tk[k] = 0.0f;
                    if (__float_as_uint(hi[k].w) != BVH_NONE) hk[k] = bvh_box(r, lo[k], hi[k], m, limit, tk[k]);
                }
            }
            // the slot that the first interested lane would enter first is walked next, the others are pushed
            unsigned long long votes[4]; float rep[4];
            int which = -1; float rep_best = __builtin_inff();
#pragma unroll
            for (int k = 0; k < 4; k++) {
                votes[k] = __ballot(hk[k]);
                rep[k] = __builtin_inff();
                if (votes[k]) {
                    int leader = __ffsll((long long)votes[k]) - 1;
                    rep[k] = fminf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(tk[k]), leader)), 3.0e38f);
                    if (which < 0 || rep[k] < rep_best) { which = k; rep_best = rep[k]; }
                }
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (votes[k] != 0ull && k != which) {
                    if (sp < pk_limit) {
                        if (lane == 0) {
                            pst[sp] = __float_as_uint(hi[k].w);
                            pst[PK_STACK + sp] = (uint32_t)votes[k];
                            pst[2 * PK_STACK + sp] = (uint32_t)(votes[k] >> 32);
                        }
                        sp++;
                    } else ok = false;
                }
            }
            if (!ok) break;
            if (which >= 0) {
                uint32_t nx = 0u; unsigned long long nm = 0ull;
#pragma unroll
                for (int k = 0; k < 4; k++) if (k == which) { nx = __float_as_uint(hi[k].w); nm = votes[k]; }
                node = nx; mask = nm;
            } else node = BVH_NONE;
        }
    }
    return ok;
}

// One ray per lane, start to finish.  ANY = false: closest triangle hit into `best` (limit =
// largest t still of interest, e.g. the sphere winner).  ANY = true: returns whether any triangle
// is hit.  `stk` points at this lane's column of the LDS stack (entries BS words apart, BS = threads per block).
template <int MODE, bool ANY, int BS = 256>
DEV bool bvh_traverse(const DBvh &B, const TRay &r, f3 o, f3 d, float limit, Best &best, uint32_t *stk,
                      uint32_t &n_nodes, uint32_t &n_tris, const float4 *__restrict__ nodes_sc = nullptr)
{
    TState st; st.node = B.root; st.sp = 0; st.limit = limit; st.found = 0u; st.overflow = 0u;
    // step scheduling weight 16: the wave takes the leaf step unless twice as many lanes wait for an inner-node step
    // (a leaf step is what ends rays and shrinks limits; measured against 8, plain majority: C2 -2 %, C3 -2.5 %)
    bvh_burst<MODE, ANY, BS>(B, r, o, d, st, best, stk, 1, n_nodes, n_tris, RT_SCHED_W, nodes_sc);
    if (st.overflow) {
        n_nodes |= 0x80000000u;                 // flag for bvh_count: this lane fell back to the linear scan
        st.found = bvh_linear<MODE, ANY>(B, o, d, best, n_tris) ? 1u : 0u;
    }
    return st.found != 0u;
}

constexpr int SPH_LDS = 64;            // spheres a block keeps in LDS

// every thread of the block must call this; returns the number of spheres held in `lds`
DEV uint32_t spheres_to_lds(const DScene &S, float4 *lds) {
    const uint32_t n_sl = min(S.n_sph, (uint32_t)SPH_LDS);
    if (threadIdx.x < n_sl) lds[threadIdx.x] = S.sph[threadIdx.x];
    __syncthreads();
    return n_sl;
}

// Workgroups are handed to the 8 XCDs in turn (workgroup b runs on XCD b mod 8), each XCD with an L2 of its own.
// Runs of XCD_RUN consecutive work items go to one XCD, the runs to the XCDs in turn: neighbouring tiles share an L2
// and the XCDs still get the same mix of cheap and expensive parts of the image.
#ifndef XCD_RUN
#define XCD_RUN 16u
#endif
DEV uint32_t xcd_block() {
    const uint32_t b = blockIdx.x, g = gridDim.x;
    const uint32_t full = g / (8u * XCD_RUN) * (8u * XCD_RUN);          // the ragged end keeps the plain order
    if (b >= full) return b;
    const uint32_t xcd = b & 7u, idx = b >> 3;                            // idx-th workgroup of this XCD
    return ((idx / XCD_RUN) * 8u + xcd) * XCD_RUN + idx % XCD_RUN;
}

// lanes with `want` get consecutive indices from the LDS counter *next (one LDS atomic per wave);
// returns the first index past the ones handed out to this wave
DEV uint32_t wave_take(bool want, uint32_t *next, uint32_t &idx) {
    unsigned long long m = __ballot(want);
    uint32_t cnt = (uint32_t)__popcll(m), start = 0;
    int lane = lane_id();
    if (m) {
        int leader = __ffsll((long long)m) - 1;
        if (lane == leader) start = atomicAdd(next, cnt);
        start = (uint32_t)__shfl((int)start, leader);
    }
    idx = start + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    return start + cnt;
}

// Closest hit of n queued rays.  A block owns `per_block` consecutive rays (bits 16.. of the last argument) and its
// lanes take them one at a time: a lane that finishes its ray writes the hit and fetches the
// next ray of the block, so waves stay populated although rays differ widely in length.
constexpr int REFILL_BELOW = 44;       // active lanes below which a wave goes back to fetch rays

template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6))) void k_bvh_closest_p(DScene S, DBvh B, const float4 *__restrict__ rays, uint32_t n,
                                                       float4 *__restrict__ hits, unsigned long long *__restrict__ counters,
                                                       int klass, int refill_below, const uint32_t *__restrict__ n_dev)
{
    __shared__ uint32_t stack[BVH_STACK][256];
    __shared__ uint32_t next;
    n = queue_fill(n, n_dev);
    const uint32_t per_block = (uint32_t)refill_below >> 16;          // rays owned by a block
    const uint32_t base = xcd_block() * per_block;
    const uint32_t end = min(n, base + per_block);
    if (threadIdx.x == 0) next = base;
    __syncthreads();
    uint32_t *stk = &stack[0][threadIdx.x];
    uint32_t n_nodes = 0, n_tris = 0, j = 0;
    uint32_t have = 0u;
    f3 o = mk(0.f, 0.f, 0.f), d = mk(0.f, 0.f, 1.f);
    TRay r = tray_make(S, B, o, d);
    TState st; st.node = BVH_NONE; st.sp = 0; st.limit = 0.f; st.found = 0u; st.overflow = 0u;
    Best tb = best_none(), sb = best_none();
    for (;;) {
        bool idle = st.node == BVH_NONE;
        if (idle && have) {
            if (st.overflow) { bvh_linear<MODE, false>(B, o, d, tb, n_tris); n_nodes |= 0x80000000u; }
            hits[j] = merge_hits<MODE>(S, tb, sb);
            have = 0u;
        }
        uint32_t idx;
        uint32_t past = wave_take(idle, &next, idx);
        if (idle && idx < end) {
            uint32_t s, aux;
            load_ray(rays, idx, o, d, s, aux);
            if (aux == 0u) {
                j = idx; have = 1u;
                sb = spheres_closest<MODE>(S, o, d);
                tb = best_none();
                r = tray_make(S, B, o, d);
                st.node = B.root; st.sp = 0; st.found = 0u; st.overflow = 0u;
                st.limit = sb.id >= 0 ? sb.t : __builtin_inff();
            }
        }
        const bool more = past < end;
        if (__ballot(st.node != BVH_NONE || have) == 0 && !more) break;
        bvh_burst<MODE, false>(B, r, o, d, st, tb, stk, more ? (refill_below & 127) : 1, n_nodes, n_tris, (refill_below >> 8) & 255);
    }
    bvh_count(counters, klass, n_nodes, n_tris, 0u);
}

// camera rays of the pixel grid into ray records; sample s = sample_index(slot, r).  Slots
// outside the frame get aux = 1 and a hit record with id -2, and are skipped by k_bvh_closest.
template <int MODE>
__global__ __launch_bounds__(256) void k_raygen(DScene S, Frame F, float4 *__restrict__ rays, float4 *__restrict__ hits)
{
    constexpr int SPP = (MODE == HIPEXEC_MODE_CL) ? 1 : 4;
    uint32_t slot = blockIdx.x * 256u + threadIdx.x;
    int x, y, lrow;
    bool active = slot_to_pixel(F, slot, x, y, lrow);
    if (slot >= F.n_slots) return;
    if (!active) { x = min(x, F.W - 1); y = min(y, F.H - 1); }
    f3 d[SPP];
    camera_rays<MODE>(S, F, x, y, d);
    f3 o = mk(S.cam_pos[0], S.cam_pos[1], S.cam_pos[2]);
#pragma unroll
    for (int r = 0; r < SPP; r++) {
        uint32_t s = sample_index<SPP>(F, slot, r);
        store_ray(rays, s, o, d[r], s, active ? 0u : 1u);
        if (!active) hits[s] = make_float4(0.f, 0.f, 0.f, __int_as_float(-2));
    }
}

// counters[8 + klass] += nodes visited, counters[11 + klass] += exact triangle tests
// (klass 0 camera rays, 1 other closest-hit rays, 2 shadow rays)
// counters[15] += rays that overflowed the traversal stack (flagged in bit 31 of n_nodes).
// Summed per wave, then per block in LDS, so a block issues at most three global atomics; every
// thread of the block must call this (it contains barriers).
DEV void bvh_count(unsigned long long *counters, int klass, uint32_t n_nodes, uint32_t n_tris, uint32_t n_rays) {
    unsigned long long a = wave_sum_u32(n_nodes & 0x7fffffffu), b = wave_sum_u32(n_tris), c = wave_sum_small(n_nodes >> 31);
    unsigned long long r = wave_sum_u32(n_rays);
    // blocks add to one of CNT_STRIPES copies of the counters (atomics on one address are serialised by the L2)
    counters += (size_t)(blockIdx.x & (CNT_STRIPES - 1u)) * CNT_STRIDE;
    if (blockDim.x == 64u) {                          // one wave: its sums are the block's
        if (threadIdx.x == 0) {
            if (a) atomicAdd(&counters[8 + klass], a);
            if (b) atomicAdd(&counters[11 + klass], b);
            if (c) atomicAdd(&counters[15], c);
            if (r) atomicAdd(&counters[14], r);
        }
        return;
    }
    __shared__ unsigned long long block_sum[4];
    if (threadIdx.x < 4) block_sum[threadIdx.x] = 0ull;
    __syncthreads();
    if ((threadIdx.x & 63u) == 0) {
        if (a) atomicAdd(&block_sum[0], a);
        if (b) atomicAdd(&block_sum[1], b);
        if (c) atomicAdd(&block_sum[2], c);
        if (r) atomicAdd(&block_sum[3], r);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (block_sum[0]) atomicAdd(&counters[8 + klass], block_sum[0]);
        if (block_sum[1]) atomicAdd(&counters[11 + klass], block_sum[1]);
        if (block_sum[2]) atomicAdd(&counters[15], block_sum[2]);
        if (block_sum[3]) atomicAdd(&counters[14], block_sum[3]);     // shadow rays traced without a queue
    }
}

// BS = threads per block.  A block's LDS (its stacks) is released only when its last wave has finished, and rays
// differ widely in length: with one wave per block no finished wave keeps memory from a waiting block.
template <int MODE, int BS>
__global__ __launch_bounds__(BS) __attribute__((amdgpu_waves_per_eu(RT_CLOSEST_WAVES))) void k_bvh_closest(DScene S, DBvh B, const float4 *__restrict__ rays, uint32_t n,
                                                     float4 *__restrict__ hits, unsigned long long *__restrict__ counters,
                                                     int klass, const uint32_t *__restrict__ n_dev, const float4 *__restrict__ nodes_sc)
{
    __shared__ uint32_t stack[BVH_STACK][BS];
    n = queue_fill(n, n_dev);
    uint32_t j = xcd_block() * (uint32_t)BS + threadIdx.x;
    bool valid = j < n;
    f3 o, d; uint32_t s, aux;
    load_ray(rays, valid ? j : 0u, o, d, s, aux);
    valid = valid && aux == 0u;
    uint32_t n_nodes = 0, n_tris = 0;
    if (valid) {
        Best sb = spheres_closest<MODE>(S, o, d);
        Best tb = best_none();
        TRay r = tray_make(S, B, o, d);
        float limit = sb.id >= 0 ? sb.t : __builtin_inff();
        bvh_traverse<MODE, false, BS>(B, r, o, d, limit, tb, &stack[0][threadIdx.x], n_nodes, n_tris, nodes_sc);
        hits[j] = merge_hits<MODE>(S, tb, sb);
    }
    bvh_count(counters, klass, n_nodes, n_tris, 0u);
}

// Closest hits of rays that arrive in coherent groups of 64 (camera rays: one 4x4-pixel quadrant with its sub-samples
// per wave): packet walk.  GEN: the kernel makes the camera rays itself (thread j = sample j: what k_raygen would have
// stored for it, from the same expressions, hence the same bits), stores them for the kernels that light and shade the
// level, and walks them at once: one launch and one pass over the ray records less per frame.
template <int MODE, bool GEN>
__global__ __launch_bounds__(256) void k_bvh_closest_pk(DScene S, DBvh B, float4 *__restrict__ rays, uint32_t n,
                                                        float4 *__restrict__ hits, unsigned long long *__restrict__ counters,
                                                        int klass, Frame F)
{
    __shared__ uint32_t pstack[4][3 * PK_STACK];
    uint32_t j = xcd_block() * 256u + threadIdx.x;
    bool valid = j < n;
    f3 o, d; uint32_t s, aux;
    if (GEN) {
        constexpr int SPP = (MODE == HIPEXEC_MODE_CL) ? 1 : 4;
        // inverse of sample_index: sample j -> pixel slot and sub-sample
        uint32_t slot = j; int r = 0;
        if (SPP == 4) {
            const uint32_t pix = j & 15u, quad = (j >> 6) & 3u, blk = j >> 8;
            r = (int)((j >> 4) & 3u);
            slot = blk * 64u + ((quad >> 1) * 4u + (pix >> 2)) * 8u + (quad & 1u) * 4u + (pix & 3u);
        }
        int x, y, lrow;
        const bool active = slot_to_pixel(F, slot, x, y, lrow);
        if (!active) { x = min(x, F.W - 1); y = min(y, F.H - 1); }
        o = mk(S.cam_pos[0], S.cam_pos[1], S.cam_pos[2]);
        d = camera_ray<MODE>(S, F, x, y, r);
        aux = active ? 0u : 1u;
        if (valid) {
            store_ray(rays, j, o, d, j, aux);
            if (!active) hits[j] = make_float4(0.f, 0.f, 0.f, __int_as_float(-2));
        }
    } else {
        load_ray(rays, valid ? j : 0u, o, d, s, aux);
    }
    valid = valid && aux == 0u;
    uint32_t n_nodes = 0, n_tris = 0, found = 0;
    Best sb = spheres_closest<MODE>(S, o, d);
    Best tb = best_none();
    TRay r = tray_make(S, B, o, d);
    float limit = sb.id >= 0 ? sb.t : __builtin_inff();
    bool ok = bvh_packet<MODE, false>(B, r, o, d, valid, limit, tb, found, pstack[threadIdx.x >> 6], n_nodes, n_tris);
    if (!ok && valid) { bvh_linear<MODE, false>(B, o, d, tb, n_tris); n_nodes |= 0x80000000u; }
    if (valid) hits[j] = merge_hits<MODE>(S, tb, sb);
    bvh_count(counters, klass, n_nodes, n_tris, 0u);
}

// occlusion of queued shadow rays: bit `lamp` of occl[s * lw + lamp / 32] is set for occluded rays
template <int MODE>
__global__ __launch_bounds__(256) void k_bvh_anyhit(DScene S, DBvh B, const float4 *__restrict__ rays, uint32_t n,
                                                    uint32_t *__restrict__ occl, uint32_t lw,
                                                    unsigned long long *__restrict__ counters)
{
    __shared__ uint32_t stack[BVH_STACK][256];
    uint32_t j = blockIdx.x * 256u + threadIdx.x;
    bool valid = j < n;
    f3 o, d; uint32_t s, lamp;
    load_ray(rays, valid ? j : 0u, o, d, s, lamp);
    uint32_t n_nodes = 0, n_tris = 0;
    if (valid) {
        bool occluded = false;
        for (uint32_t k = 0; k < S.n_sph && !occluded; k++) {
            float4 sp = S.sph[k];
            float t;
            occluded = sph_exact<MODE>(o, d, xyz(sp), sp.w, t);
        }
        if (!occluded) {
            TRay r = tray_make(S, B, o, d);
            Best none = best_none();
            occluded = bvh_traverse<MODE, true>(B, r, o, d, __builtin_inff(), none, &stack[0][threadIdx.x], n_nodes, n_tris);
        }
        if (occluded) atomicOr(&occl[(size_t)s * lw + (lamp >> 5)], 1u << (lamp & 31u));
    }
    bvh_count(counters, 2, n_nodes, n_tris, 0u);
}

// ---------------------------------------------------------------------------------------
// k_bvh_shadow: shadow rays of one level without a global queue.  A block owns `per_block` <= 256
// rays of the level (fewer for small levels, so that the launch still covers the chip).  Per group of SH_LAMPS lamps it evaluates the facing tests (ray_tracer.cpp:174-182 /
// ray_tracer_cl.cpp:138-143), compacts the (sample, lamp) pairs that cast a shadow ray into an
// LDS list in lamp-major order (ballot + popcount, no global atomics), and then traces the
// list 256 entries at a time, so waves stay full and their rays point at the same lamp from
// neighbouring surface points.  Occluded pairs set bit `lamp` of occl[s * lw + lamp / 32].
// counters[14] += shadow rays traced.
// ---------------------------------------------------------------------------------------
constexpr int SH_LAMPS = 4;

template <int MODE, bool FUSED>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5))) void k_bvh_shadow(DScene S, DBvh B, const float4 *__restrict__ rays,
                                                    const float4 *__restrict__ hits, uint32_t n,
                                                    uint32_t *__restrict__ occl, uint32_t lw,
                                                    unsigned long long *__restrict__ counters, uint32_t per_block,
                                                    const float4 *__restrict__ nodes_sc, float4 *__restrict__ local, int spawn,
                                                    float4 *__restrict__ next_rays, uint32_t *__restrict__ next_count,
                                                    const uint32_t *__restrict__ n_dev)
{
    n = queue_fill(n, n_dev);
    __shared__ uint32_t occ_lds[FUSED ? 256 : 1];          // FUSED (at most 32 lamps): the block's occlusion bits stay here
    __shared__ uint32_t stack[BVH_STACK][256];
    __shared__ float Px[256], Py[256], Pz[256];
    __shared__ uint32_t sample_of[256];
    __shared__ uint16_t list[SH_LAMPS * 256];
    __shared__ uint32_t wcnt[SH_LAMPS * 4 + 1];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    uint32_t j = xcd_block() * per_block + tid;
    bool valid = tid < per_block && j < n;
    f3 o, d; uint32_t s, aux;
    load_ray(rays, valid ? j : 0u, o, d, s, aux);
    float4 h = hits[valid ? j : 0u];
    bool hit = valid && __float_as_int(h.w) >= 0;
    Surf sf;
    if (hit) sf = surface(S, o, d, h, false);
    else { sf.P = mk(0.f, 0.f, 0.f); sf.N = mk(0.f, 0.f, 1.f); }
    Px[tid] = sf.P.x; Py[tid] = sf.P.y; Pz[tid] = sf.P.z;
    sample_of[tid] = s;
    if (FUSED) occ_lds[tid] = 0u;
    const float dotWithDir = dot(d, sf.N);
    uint32_t n_nodes = 0, n_tris = 0, n_rays = 0;
    for (uint32_t l0 = 0; l0 < S.n_lamp; l0 += SH_LAMPS) {
        unsigned long long votes[SH_LAMPS];
#pragma unroll
        for (int li = 0; li < SH_LAMPS; li++) {
            bool cast = false;
            if (l0 + li < S.n_lamp) {
                float4 lp = S.lamp[l0 + li];
                f3 toLamp = xyz(lp) - sf.P;
                cast = hit && casts_shadow_ray<MODE>(dotWithDir, dot(toLamp, sf.N));
            }
            votes[li] = __ballot(cast);
            if (lane == 0) wcnt[li * 4 + w] = (uint32_t)__popcll(votes[li]);
        }
        __syncthreads();
        if (tid < 64) {                     // exclusive prefix over (lamp, wave) in lamp-major order
            uint32_t own = tid < SH_LAMPS * 4 ? wcnt[tid] : 0u, inc = own;
            for (int off = 1; off < SH_LAMPS * 4; off <<= 1) {
                uint32_t up = (uint32_t)__shfl_up((int)inc, off);
                if ((int)lane >= off) inc += up;
            }
            uint32_t total = (uint32_t)__shfl((int)inc, SH_LAMPS * 4 - 1);
            if (tid < SH_LAMPS * 4) wcnt[tid] = inc - own;
            if (tid == 0) wcnt[SH_LAMPS * 4] = total;
        }
        __syncthreads();
        const uint32_t total = wcnt[SH_LAMPS * 4];
#pragma unroll
        for (int li = 0; li < SH_LAMPS; li++) {
            if ((votes[li] >> lane) & 1ull) {
                uint32_t pos = wcnt[li * 4 + w] + (uint32_t)__popcll(votes[li] & ((1ull << lane) - 1ull));
                list[pos] = (uint16_t)(tid | ((uint32_t)li << 8));
            }
        }
        __syncthreads();
        uint32_t my_bits = 0;               // not used: results go out through atomics on the owner's word
        (void)my_bits;
        for (uint32_t base = 0; base < total; base += 256u) {
            uint32_t e = base + tid;
            if (e < total) {
                uint32_t entry = list[e];
                uint32_t t2 = entry & 255u, l = l0 + (entry >> 8);
                f3 P = mk(Px[t2], Py[t2], Pz[t2]);
                float4 lp = S.lamp[l];
                f3 toLamp = xyz(lp) - P;
                bool occluded = false;
                for (uint32_t k = 0; k < S.n_sph && !occluded; k++) {
                    float4 sp = S.sph[k];
                    float t;
                    occluded = sph_exact<MODE>(P, toLamp, xyz(sp), sp.w, t);
                }
                if (!occluded) {
                    TRay r = tray_make(S, B, P, toLamp);
                    Best none = best_none();
                    occluded = bvh_traverse<MODE, true>(B, r, P, toLamp, __builtin_inff(), none, &stack[0][tid], n_nodes, n_tris, nodes_sc);
                }
                if (occluded) {
                    if (FUSED) atomicOr(&occ_lds[t2], 1u << l);
                    else atomicOr(&occl[(size_t)sample_of[t2] * lw + (l >> 5)], 1u << (l & 31u));
                }
                n_rays++;
            }
        }
        __syncthreads();
    }
    if (FUSED) {
        // shading of the block's samples right here: the occlusion bits never leave LDS and k_shade's launch is saved
        // (ray and hit are read again rather than kept in registers through the traversal)
        bool bounce = false;
        f3 bo = mk(0.f, 0.f, 0.f), bd = mk(0.f, 0.f, 1.f);
        if (valid) {
            load_ray(rays, j, o, d, s, aux);
            float4 hh = hits[j];
            if (__float_as_int(hh.w) >= -1)
                local[s] = shade_sample<MODE>(S, o, d, hh, &occ_lds[tid], spawn != 0, bounce, bo, bd);
        }
        if (spawn && next_rays) {
            __shared__ uint32_t append_scratch[5];
            uint32_t pos = block_append(bounce, next_count, append_scratch);
            if (bounce) store_ray(next_rays, pos, bo, bd, s, 0u);
        }
    }
    bvh_count(counters, 2, n_nodes, n_tris, n_rays);
}

// ---------------------------------------------------------------------------------------
// k_bvh_shadow_direct: the same work without the block's list.  A lane keeps its sample and traces that sample's ray
// to one lamp after the other; lanes whose sample casts no ray to the lamp at hand sit that lamp out.  No barriers and
// no LDS besides the stacks, but the waves are only as full as the fraction of samples that face the lamp.
// ---------------------------------------------------------------------------------------
// AO: the kernel also traces the occlusion rays of HIPEXEC_FLAG_AO (a variant of its own, so that the default kernel does
// not carry that loop's registers)
// WAVES: waves per SIMD the kernel is compiled for: RT_SHADOW_WAVES_MANY = 7 (72 VGPRs, nothing spilled since the generic
// pow of the shading is a call; seven blocks' stacks of 21 entries fit a CU's LDS) for the kernels of whole frames (7
// against 6: C2 -2.3 %, C3 -2.2 %, C5 -3.9 % per frame, C4 with its one lamp -2 % against 5; 8 waves with an 18-entry
// stack spill and lose all of it again; while the shading still spilled the pow's constants, 7 had measured +-0); the
// default of 5 (96 VGPRs) remains for the ambient-occlusion variant and the HIPEXEC_SHADOW_WAVES6_FROM knob.
template <int MODE, bool FUSED, int BS, bool AO = false, int WAVES = RT_SHADOW_WAVES>
__global__ __launch_bounds__(BS) __attribute__((amdgpu_waves_per_eu(WAVES))) void k_bvh_shadow_direct(DScene S, DBvh B, const float4 *__restrict__ rays,
                                                    const float4 *__restrict__ hits, uint32_t n,
                                                    uint32_t *__restrict__ occl, uint32_t lw,
                                                    unsigned long long *__restrict__ counters, uint32_t per_block,
                                                    const float4 *__restrict__ nodes_sc, float4 *__restrict__ local, int spawn,
                                                    float4 *__restrict__ next_rays, uint32_t *__restrict__ next_count,
                                                    const uint32_t *__restrict__ n_dev)
{
    n = queue_fill(n, n_dev);
    __shared__ uint32_t stack[BVH_STACK][BS];
    // The spheres are tested one after the other for every shadow ray, each test behind a load of its own: from LDS
    // that load takes a tenth of the time it takes from the vector cache (first SPH_LDS spheres; the rest from memory).
    // C5 (16 spheres) -2.5 %; the closest-hit kernels test the spheres once per ray and gain nothing from the same copy.
    __shared__ float4 sph_lds[SPH_LDS];
    const uint32_t tid = threadIdx.x;
    const uint32_t n_sl = spheres_to_lds(S, sph_lds);
    uint32_t j = xcd_block() * per_block + tid;
    bool valid = tid < per_block && j < n;
    f3 o, d; uint32_t s, aux;
    load_ray(rays, valid ? j : 0u, o, d, s, aux);
    float4 h = hits[valid ? j : 0u];
    bool hit = valid && __float_as_int(h.w) >= 0;
    Surf sf;
    if (hit) sf = surface(S, o, d, h, false);
    else { sf.P = mk(0.f, 0.f, 0.f); sf.N = mk(0.f, 0.f, 1.f); }
    const f3 P = sf.P;
    const float dotWithDir = dot(d, sf.N);
    uint32_t n_nodes = 0, n_tris = 0, n_rays = 0;
    uint32_t occ_bits = 0u;                     // FUSED: at most 32 lamps
    for (uint32_t l = 0; l < S.n_lamp; l++) {
        float4 lp = S.lamp[l];
        f3 toLamp = xyz(lp) - P;
        const bool cast = hit && casts_shadow_ray<MODE>(dotWithDir, dot(toLamp, sf.N));
        if (cast) {
            bool occluded = false;
            for (uint32_t k = 0; k < n_sl && !occluded; k++) {
                float4 sp = sph_lds[k];
                float t;
                occluded = sph_exact<MODE>(P, toLamp, xyz(sp), sp.w, t);
            }
            for (uint32_t k = n_sl; k < S.n_sph && !occluded; k++) {
                float4 sp = S.sph[k];
                float t;
                occluded = sph_exact<MODE>(P, toLamp, xyz(sp), sp.w, t);
            }
            if (!occluded) {
                TRay r = tray_make(S, B, P, toLamp);
                Best none = best_none();
                occluded = bvh_traverse<MODE, true, BS>(B, r, P, toLamp, __builtin_inff(), none, &stack[0][tid], n_nodes, n_tris, nodes_sc);
            }
            if (occluded) {
                if (FUSED) occ_bits |= 1u << l;
                else atomicOr(&occl[(size_t)s * lw + (l >> 5)], 1u << (l & 31u));
            }
            n_rays++;
        }
    }
    // HIPEXEC_FLAG_AO: the occlusion rays of computeAmbientOcclusion (ray_tracer.cpp:406-424), one after the other like the
    // lamp rays; the sum is formed in the reference's order
    float ao = 0.0f;
    if (AO && FUSED && MODE == HIPEXEC_MODE_CPU && S.ao) {
        const f3 realNorm = dotWithDir < 0.0f ? sf.N : -sf.N;
        for (uint32_t i = 0; i < (uint32_t)HIPEXEC_AO_RAYS; i++) {
            if (hit) {
                const f3 rd = ao_ray(P, realNorm, i);
                bool occluded = false;
                for (uint32_t k = 0; k < n_sl && !occluded; k++) { float4 sp = sph_lds[k]; float t; occluded = sph_exact<MODE>(P, rd, xyz(sp), sp.w, t); }
                for (uint32_t k = n_sl; k < S.n_sph && !occluded; k++) { float4 sp = S.sph[k]; float t; occluded = sph_exact<MODE>(P, rd, xyz(sp), sp.w, t); }
                if (!occluded) {
                    TRay r = tray_make(S, B, P, rd);
                    Best none = best_none();
                    occluded = bvh_traverse<MODE, true, BS>(B, r, P, rd, __builtin_inff(), none, &stack[0][tid], n_nodes, n_tris, nodes_sc);
                }
                if (!occluded) ao += S.ambient_factor;
                n_rays++;
            }
        }
        ao = ao / (float)HIPEXEC_AO_RAYS;
    }
    if (FUSED) {
        bool bounce = false;
        f3 bo = mk(0.f, 0.f, 0.f), bd = mk(0.f, 0.f, 1.f);
        if (valid) {
            load_ray(rays, j, o, d, s, aux);
            float4 hh = hits[j];
            if (__float_as_int(hh.w) >= -1)
                local[s] = shade_sample<MODE>(S, o, d, hh, &occ_bits, spawn != 0, bounce, bo, bd, ao);
        }
        if (spawn && next_rays) {
            __shared__ uint32_t append_scratch[5];
            uint32_t pos = block_append(bounce, next_count, append_scratch);
            if (bounce) store_ray(next_rays, pos, bo, bd, s, 0u);
        }
    }
    bvh_count(counters, 2, n_nodes, n_tris, n_rays);
}

// ---------------------------------------------------------------------------------------
// k_bvh_tail: the remaining recursion levels of a small ray queue in one launch.  A lane follows its
// ray from level `level0` to the end of its path (closest hit, shadow rays, shading, bounce, and so
// on), writing local[level][s] exactly as k_shade does.  Used when the queue of a level is so small that
// every separate launch would cost its floor (the longest ray) while most of the chip idles: the floors of
// all remaining levels then collapse into one.  Same device functions, same operation order per
// sample, hence the same bits as the level-by-level kernels.
// counters[16] += bounce rays generated here (levels above level0).
// ---------------------------------------------------------------------------------------
template <int MODE, int BS>
__global__ __launch_bounds__(BS) __attribute__((amdgpu_waves_per_eu(5))) void k_bvh_tail(DScene S, DBvh B, const float4 *__restrict__ rays, uint32_t n, int level0,
                                                  int depth, float4 *__restrict__ local, size_t level_stride,
                                                  unsigned long long *__restrict__ counters, const uint32_t *__restrict__ n_dev)
{
    __shared__ uint32_t stack[BVH_STACK][BS];
    n = queue_fill(n, n_dev);
    uint32_t *stk = &stack[0][threadIdx.x];
    uint32_t j = blockIdx.x * (uint32_t)BS + threadIdx.x;
    bool alive = j < n;
    f3 o, d; uint32_t s, aux;
    load_ray(rays, alive ? j : 0u, o, d, s, aux);
    uint32_t nodes_c = 0, tris_c = 0, nodes_s = 0, tris_s = 0, n_shadow = 0, n_bounce = 0;
    for (int level = level0; alive; level++) {
        Best sb = spheres_closest<MODE>(S, o, d);
        Best tb = best_none();
        TRay r = tray_make(S, B, o, d);
        bvh_traverse<MODE, false, BS>(B, r, o, d, sb.id >= 0 ? sb.t : __builtin_inff(), tb, stk, nodes_c, tris_c);
        float4 h = merge_hits<MODE>(S, tb, sb);
        f3 ret; float w = -1.0f;
        if (__float_as_int(h.w) < 0) {
            f3 hz = mk(S.horizon[0], S.horizon[1], S.horizon[2]);
            ret = (MODE == HIPEXEC_MODE_CL) ? (mk(0.f, 0.f, 0.f) + hz) : hz;
            alive = false;
        } else {
            Surf sf = surface(S, o, d, h, true);
            ret = (MODE == HIPEXEC_MODE_CL) ? mk(0.f, 0.f, 0.f) : mk(S.ambient[0], S.ambient[1], S.ambient[2]);
            for (uint32_t l = 0; l < S.n_lamp; l++) {
                float4 lp = S.lamp[l];
                f3 toLamp = xyz(lp) - sf.P;
                float dotWithLamp = dot(toLamp, sf.N);
                float dotWithDir = dot(d, sf.N);
                if (!casts_shadow_ray<MODE>(dotWithDir, dotWithLamp)) continue;
                n_shadow++;
                bool occluded = false;
                for (uint32_t k = 0; k < S.n_sph && !occluded; k++) {
                    float4 sp = S.sph[k];
                    float t;
                    occluded = sph_exact<MODE>(sf.P, toLamp, xyz(sp), sp.w, t);
                }
                if (!occluded) {
                    TRay rs = tray_make(S, B, sf.P, toLamp);
                    Best none = best_none();
                    occluded = bvh_traverse<MODE, true, BS>(B, rs, sf.P, toLamp, __builtin_inff(), none, stk, nodes_s, tris_s);
                }
                if (occluded) continue;
                ret = add_lamp<MODE>(ret, sf, d, toLamp, lp.w, S.lamp_dist[l]);
            }
            f3 bd;
            if (MODE == HIPEXEC_MODE_CPU && level < depth && secondary_ray(S, sf.mat, sf.kr, d, sf.N, w, bd)) {
                o = sf.P; d = bd;
                n_bounce++;
            } else { w = -1.0f; alive = false; }
        }
        local[(size_t)level * level_stride + s] = make_float4(ret.x, ret.y, ret.z, w);
    }
    bvh_count(counters, 1, nodes_c, tris_c, 0u);
    __syncthreads();                        // bvh_count's LDS sums are reused by the second call
    bvh_count(counters, 2, nodes_s, tris_s, n_shadow);
    unsigned long long nb = wave_sum_u64(n_bounce);
    if ((threadIdx.x & 63u) == 0 && nb) atomicAdd(&counters[(size_t)(blockIdx.x & (CNT_STRIPES - 1u)) * CNT_STRIDE + 16], nb);
}

}  // namespace rt
