// rt_bvh_host.hpp — topology of the hierarchy from a surface-area-heuristic build on the host.
//
// The device builds its hierarchy in under a millisecond from Morton order (rt_bvh.hpp), which is what an executor
// that lives for one frame needs (ray_tracer_cl.cpp:14).  A scene that is kept can pay for a better tree once: a top-down
// binned surface-area build groups the triangles by what their boxes cost, not by a fixed grid (5-10 % of a frame on
// scenes of boxes and walls; the device build, which took over what this one taught - box centres, tight grid, orientation
// classes - is level with it on the regular terrains).  Only the TOPOLOGY is made here: the
// order of the triangles and the binary tree over that order, in the numbering of k_bvh_split, so that the device's
// refit and emit kernels, the 4-wide records and the traversal are the same for both builds.  The hierarchy decides
// which exact tests run, never their outcome, so frames are bit-identical with either tree.
//
// Plain C++ (no HIP types): tests/test_bvh_host.py compiles it alone and checks the invariants.
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

namespace rt {

struct HostTopology {
    std::vector<unsigned long long> keys;   // [n]      k-th leaf: (k << 32) | triangle index, the form k_bvh_leaves reads
    std::vector<uint32_t> child;            // [2(n-1)] inner nodes 0 .. n-2 (root 0); leaf k is node n-1+k
    std::vector<uint32_t> parent;           // [2n-1]   0xffffffff for the root
    std::vector<uint32_t> range;            // [2(n-1)] first and last leaf of an inner node (its leaves are contiguous)
    std::vector<uint32_t> top_list;         // inner nodes whose leaves reach across blocks of `refit_block` leaves, any order
};

namespace sah_detail {

constexpr int BINS = 8;                     // 8, 16, 32 and 64 bins measure the same frame times, and so does a full sweep
                                            // (every position on every axis) for the ranges of up to 64 leaves
constexpr uint32_t SHARE_ABOVE = 4096;      // a thread hands one half of a range this large to the other threads

// {lo.x, lo.y, lo.z, -, -hi.x, -hi.y, -hi.z, -}: with the upper corner negated a union is one minimum over eight floats,
// which the compiler turns into two vector instructions (the cuts spend their time growing bin boxes)
struct Box {
    float v[8];
    void reset() { for (int k = 0; k < 8; k++) v[k] = 3.0e38f; }
    void grow(const Box &b) { for (int k = 0; k < 8; k++) v[k] = std::min(v[k], b.v[k]); }
    float lo(int k) const { return v[k]; }
    float hi(int k) const { return -v[4 + k]; }
    float area() const {
        float d[3];
        for (int k = 0; k < 3; k++) d[k] = std::max(0.f, hi(k) - lo(k));
        return d[0] * d[1] + d[1] * d[2] + d[2] * d[0];
    }
};

// Orientation class of a triangle record, see bvh_normal_class in rt_bvh.hpp (the device build's leading code bits)
#ifndef RT_BVH_CELL_BITS
#define RT_BVH_CELL_BITS 6
#endif
constexpr int HOST_CELL_BITS = RT_BVH_CELL_BITS;
constexpr uint32_t CELL_WORDS = (1u << (2 * HOST_CELL_BITS)) / 32u;       // words of a bit mask over the cells

// Cell of a normal direction (any length, either sign) in the octahedral map, 0xffffffff for a zero or non-finite vector
inline uint32_t normal_cell(float cx, float cy, float cz) {
    if (cx * 0.267f + cy * 0.535f + cz * 0.802f < 0.f) { cx = -cx; cy = -cy; cz = -cz; }
    const float l1 = std::fabs(cx) + std::fabs(cy) + std::fabs(cz);
    if (!(l1 > 0.f) || !(l1 < 3.0e38f)) return 0xffffffffu;
    float u = cx / l1, v = cy / l1;
    if (cz < 0.f) { const float uu = (1.0f - std::fabs(v)) * (u < 0.f ? -1.0f : 1.0f), vv = (1.0f - std::fabs(u)) * (v < 0.f ? -1.0f : 1.0f); u = uu; v = vv; }
    const float half = (float)((1 << HOST_CELL_BITS) / 2 - 1);
    const uint32_t qu = (uint32_t)std::min(2.0f * half, std::max(0.0f, (u + 1.0f) * half + 0.5f)), qv = (uint32_t)std::min(2.0f * half, std::max(0.0f, (v + 1.0f) * half + 0.5f));
    uint32_t bin = 0;
    for (int k = 0; k < HOST_CELL_BITS; k++) bin |= (((qu >> k) & 1u) << (2 * k)) | (((qv >> k) & 1u) << (2 * k + 1));
    return bin;
}

// Cells that hold at least a fifth of the scene's triangle area: planes tiled finely (a ground of small quads, the walls
// of many aligned boxes).  Their triangles are parted by orientation whatever their size; a terrain, whose normals spread
// over many cells, has none (at a twentieth the gentle terrain of C4 had, and lost 25 %).  e1 / e2: [n][stride] floats, the first three of each the edge.
inline void popular_cells(size_t n, const float *e1, const float *e2, size_t stride, uint32_t *mask /* [CELL_WORDS] */) {
    std::vector<double> area((size_t)CELL_WORDS * 32u, 0.0);
    double total = 0.0;
    const size_t step = std::max<size_t>(1, n / 16384);                    // a sample is enough for shares of a fifth
    for (size_t i = 0; i < n; i += step) {
        const float *a = e1 + i * stride, *b = e2 + i * stride;
        const float cx = b[1] * a[2] - b[2] * a[1], cy = b[2] * a[0] - b[0] * a[2], cz = b[0] * a[1] - b[1] * a[0];      // cross(e2, e1)
        const uint32_t cell = normal_cell(cx, cy, cz);
        if (cell == 0xffffffffu) continue;
        const double w = std::sqrt((double)cx * cx + (double)cy * cy + (double)cz * cz);
        if (!(w < 1.0e300)) continue;
        area[cell] += w; total += w;
    }
    for (uint32_t w = 0; w < CELL_WORDS; w++) mask[w] = 0u;
    for (uint32_t c = 0; c < CELL_WORDS * 32u; c++) if (total > 0.0 && area[c] >= 0.2 * total) mask[c >> 5] |= 1u << (c & 31u);
}

// Orientation class of a triangle record, see bvh_normal_class in rt_bvh.hpp (the device build's leading code bits)
inline uint32_t normal_class(const float *r, float big_thr, const uint32_t *popular) {
    const uint32_t cell = normal_cell(r[3], r[7], r[11]);
    if (cell == 0xffffffffu) return 0u;
    const float e1 = std::sqrt(r[4] * r[4] + r[5] * r[5] + r[6] * r[6]), e2 = std::sqrt(r[8] * r[8] + r[9] * r[9] + r[10] * r[10]);
    if (!(e1 * e2 > big_thr) && !(popular && ((popular[cell >> 5] >> (cell & 31u)) & 1u))) return 0u;
    return 1u + cell;
}

struct Job { uint32_t a, b, parent; int side; };          // leaves [a, b), b - a >= 2; the inner node they become hangs on parent's side

struct Builder {
    uint32_t n, refit_block;
    struct Prim { Box box; uint32_t tri, cls; };         // cls: the axis the triangle's normal is closest to (0..2)
    std::vector<Prim> prim;                 // the k-th leaf's triangle and its box; the cuts reorder the records themselves,
                                            // so every pass over a range reads consecutive memory
    HostTopology *out;
    uint32_t root_split = 0;                // split position of the root (set by the first job, before any other runs)
    // shared work list of the large ranges
    std::mutex m;
    std::condition_variable cv;
    std::deque<Job> shared;
    uint32_t busy = 0;

    // An inner node is named after the last leaf of its left half; the n-1 inner nodes of a binary tree over n leaves use
    // every position 0 .. n-2 once.  The root has to be node 0, so it trades names with the node that splits after leaf 0.
    uint32_t name(uint32_t split, bool is_root) const { return is_root ? 0u : (split == 0u ? root_split : split); }

    // Where to cut the leaves [a, b): the plane between two of BINS bins of the centres' extent, on the axis and at the
    // position where (area of the left boxes) x (count) + (area of the right boxes) x (count) is smallest; the middle
    // of the range where the centres coincide.  Reorders prim[a, b) and returns the first leaf of the right half.
    uint32_t split(uint32_t a, uint32_t b) {
        // Half of all ranges hold two or three leaves, and the bins' fixed cost would dominate the build: two leaves are
        // cut between them, of three the one goes alone whose leaving gives the smallest cost (the same criterion, exact)
        if (b - a == 2) return a + 1;
        // Orientation classes are parted first (they end up as subtrees under the root): a node's normal cone, which
        // decides whether a ray may use the tight enlargement of the boxes, is only as narrow as the normals below it
        {
            uint32_t cmin = 0xffffffffu, cmax = 0;
            for (uint32_t k = a; k < b; k++) { cmin = std::min(cmin, prim[k].cls); cmax = std::max(cmax, prim[k].cls); }
            if (cmin != cmax) {
                // halved by class number (neighbouring directions have neighbouring numbers), so the classes form a balanced tree
                const uint32_t cut = cmin + (cmax - cmin) / 2;
                auto it = std::partition(prim.begin() + a, prim.begin() + b, [&](const Prim &t) { return t.cls <= cut; });
                return (uint32_t)(it - prim.begin());
            }
        }
        if (b - a == 3) {
            float cost[3];
            for (int alone = 0; alone < 3; alone++) {
                Box pair = prim[a + (alone + 1) % 3].box; pair.grow(prim[a + (alone + 2) % 3].box);
                cost[alone] = prim[a + alone].box.area() + 2.0f * pair.area();
            }
            const int alone = cost[0] <= cost[1] ? (cost[0] <= cost[2] ? 0 : 2) : (cost[1] <= cost[2] ? 1 : 2);
            std::swap(prim[a], prim[a + alone]);
            return a + 1;
        }
        float clo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, chi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
        for (uint32_t k = a; k < b; k++) {
            const Box &t = prim[k].box;
            for (int ax = 0; ax < 3; ax++) { const float c = centre(t, ax); clo[ax] = std::min(clo[ax], c); chi[ax] = std::max(chi[ax], c); }
        }
        float sc[3]; bool use[3];
        for (int ax = 0; ax < 3; ax++) {
            const float ext = chi[ax] - clo[ax];
            use[ax] = ext > 0.f && ext < 3.0e38f;
            sc[ax] = use[ax] ? BINS / ext : 0.f;
        }
        Box bb[3][BINS]; uint32_t bc[3][BINS];
        for (int ax = 0; ax < 3; ax++) for (int i = 0; i < BINS; i++) { bb[ax][i].reset(); bc[ax][i] = 0; }
        for (uint32_t k = a; k < b; k++) {
            const Box &t = prim[k].box;
            for (int ax = 0; ax < 3; ax++) {
                if (!use[ax]) continue;
                const int i = bin(centre(t, ax), clo[ax], sc[ax]);
                bb[ax][i].grow(t); bc[ax][i]++;
            }
        }
        int best_ax = -1, best_bin = 0; float best = 3.0e38f;
        for (int ax = 0; ax < 3; ax++) {
            if (!use[ax]) continue;
            float ra[BINS]; uint32_t rc[BINS];
            Box acc; acc.reset(); uint32_t c = 0;
            for (int i = BINS - 1; i >= 1; i--) { acc.grow(bb[ax][i]); c += bc[ax][i]; ra[i] = acc.area(); rc[i] = c; }
            acc.reset(); c = 0;
            for (int i = 0; i < BINS - 1; i++) {
                acc.grow(bb[ax][i]); c += bc[ax][i];
                if (c == 0 || rc[i + 1] == 0) continue;
                const float cost = acc.area() * (float)c + ra[i + 1] * (float)rc[i + 1];
                if (cost < best) { best = cost; best_ax = ax; best_bin = i; }
            }
        }
        uint32_t mid = a + (b - a) / 2;
        if (best_ax >= 0) {
            const float lo = clo[best_ax], s = sc[best_ax];
            const int ax = best_ax, cut = best_bin;
            auto it = std::partition(prim.begin() + a, prim.begin() + b, [&](const Prim &t) { return bin(centre(t.box, ax), lo, s) <= cut; });
            const uint32_t m2 = (uint32_t)(it - prim.begin());
            if (m2 > a && m2 < b) mid = m2;
        }
        return mid;
    }
    static float centre(const Box &t, int ax) { return 0.5f * t.lo(ax) + 0.5f * t.hi(ax); }
    static int bin(float c, float lo, float s) {
        const float f = (c - lo) * s;
        return f > 0.f ? (f < (float)(BINS - 1) ? (int)f : BINS - 1) : 0;       // NaN -> 0
    }

    // One range: cut it, write its inner node, and return the halves that are inner nodes themselves.
    int node(const Job &j, Job next[2], std::vector<uint32_t> &top) {
        const uint32_t mid = split(j.a, j.b);
        const bool is_root = j.parent == 0xffffffffu;
        if (is_root) root_split = mid - 1;
        const uint32_t i = name(mid - 1, is_root);
        out->range[2 * (size_t)i] = j.a; out->range[2 * (size_t)i + 1] = j.b - 1;
        out->parent[i] = j.parent;
        if (!is_root) out->child[2 * (size_t)j.parent + j.side] = i;
        if (j.a / refit_block != (j.b - 1) / refit_block) top.push_back(i);
        const uint32_t half[2][2] = {{j.a, mid}, {mid, j.b}};
        int made = 0;
        for (int side = 0; side < 2; side++) {
            if (half[side][1] - half[side][0] == 1) {
                const uint32_t leaf = (n - 1) + half[side][0];
                out->child[2 * (size_t)i + side] = leaf; out->parent[leaf] = i;
            }
            else next[made++] = Job{half[side][0], half[side][1], i, side};
        }
        return made;
    }

    void worker() {
        std::vector<uint32_t> top;
        std::vector<Job> local;
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return !shared.empty() || busy == 0; });
                if (shared.empty()) break;                              // nothing queued and nobody who could queue more
                j = shared.front(); shared.pop_front();
                busy++;
            }
            local.push_back(j);
            while (!local.empty()) {
                const Job cur = local.back(); local.pop_back();
                Job next[2];
                const int made = node(cur, next, top);
                for (int k = 0; k < made; k++) {
                    if (k == 1 && next[k].b - next[k].a > SHARE_ABOVE) {
                        { std::lock_guard<std::mutex> lk(m); shared.push_back(next[k]); }
                        cv.notify_one();
                    }
                    else local.push_back(next[k]);
                }
            }
            {
                std::lock_guard<std::mutex> lk(m);
                busy--;
                out->top_list.insert(out->top_list.end(), top.begin(), top.end());
            }
            top.clear();
            cv.notify_all();
        }
    }
};

}  // namespace sah_detail

// `rec`: 12 floats per triangle, {p.xyz, c.x}{e1.xyz, c.y}{e2.xyz, c.z} with c = cross(e2, e1) (the device's exact-test
// records); n >= 2.
// Coordinates are finite (a scene with others gets no hierarchy at all).
inline void sah_topology(uint32_t n, const float *rec, uint32_t refit_block, int threads, HostTopology &out, int class_mode = 2,
                         const uint32_t *popular = nullptr)
{
    using namespace sah_detail;
    Builder b;
    b.n = n; b.refit_block = refit_block; b.out = &out;
    b.prim.resize(n);
    float big_thr = 3.0e38f;
    std::vector<uint32_t> own_mask;
    if (class_mode == 2) {
        // (the device build's rule: large from sqrt(|e1||e2|) = 16 / scene size on)
        float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
        for (uint32_t i = 0; i < n; i++) for (int k = 0; k < 3; k++) {
            const float *r = rec + 12 * (size_t)i;
            const float v0 = r[k], v1 = r[k] + r[4 + k], v2 = r[k] + r[8 + k];
            lo[k] = std::min(lo[k], std::min(v0, std::min(v1, v2))); hi[k] = std::max(hi[k], std::max(v0, std::max(v1, v2)));
        }
        const float emax = std::max(hi[0] - lo[0], std::max(hi[1] - lo[1], hi[2] - lo[2]));
        if (emax > 0.f) big_thr = 256.0f / (emax * emax);
        if (!popular) { own_mask.resize(CELL_WORDS); popular_cells(n, rec + 4, rec + 8, 12, own_mask.data()); popular = own_mask.data(); }
    }
    for (uint32_t i = 0; i < n; i++) {
        const float *r = rec + 12 * (size_t)i;
        for (int k = 0; k < 3; k++) {
            const float v0 = r[k], v1 = r[k] + r[4 + k], v2 = r[k] + r[8 + k];
            b.prim[i].box.v[k] = std::min(v0, std::min(v1, v2)); b.prim[i].box.v[4 + k] = -std::max(v0, std::max(v1, v2));
        }
        b.prim[i].box.v[3] = 0.f; b.prim[i].box.v[7] = 0.f;
        b.prim[i].tri = i;
        const float nx = std::fabs(r[3]), ny = std::fabs(r[7]), nz = std::fabs(r[11]);        // cross(e2, e1), as the device's keys read it
        if (class_mode == 0) b.prim[i].cls = 0u;
        else if (class_mode == 1) b.prim[i].cls = (nx >= ny && nx >= nz) ? 0u : (ny >= nz ? 1u : 2u);
        else b.prim[i].cls = normal_class(r, big_thr, popular);
    }
    out.child.assign(2 * (size_t)(n - 1), 0u); out.range.assign(2 * (size_t)(n - 1), 0u);
    out.parent.assign(2 * (size_t)n - 1, 0xffffffffu); out.top_list.clear();
    // the root first and alone: the names of all other nodes depend on where it splits
    std::vector<uint32_t> top;
    Job next[2];
    const int made = b.node(Job{0u, n, 0xffffffffu, 0}, next, top);
    out.top_list = top;
    for (int k = 0; k < made; k++) b.shared.push_back(next[k]);
    threads = std::max(1, std::min(threads, n < 16384u ? 1 : 64));
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; t++) pool.emplace_back([&b] { b.worker(); });
    b.worker();
    for (auto &t : pool) t.join();
    out.keys.resize(n);
    for (uint32_t k = 0; k < n; k++) out.keys[k] = ((unsigned long long)k << 32) | b.prim[k].tri;
}

}  // namespace rt
