// Invariants of the host-side surface-area topology (raytracer_amd/csrc/rt_bvh_host.hpp), checked for random soups, a
// grid, coincident triangles and tiny inputs, with one and several threads:
//   * every triangle is the leaf of exactly one position; keys carry (position << 32) | triangle
//   * inner nodes 0 .. n-2 are all used, the root is 0, parent and child agree
//   * an inner node's range is the concatenation of its children's ranges (left first)
//   * top_list holds exactly the inner nodes whose range crosses a block boundary
//   * the tree does not depend on the number of threads
//   * triangles of one orientation class (cell of the normal, large triangles only) are contiguous in the leaf order
// usage: bvh_host_check [n_for_timing]   -> prints "ok" and, with an argument, the build time of a grid of that size
#include "rt_bvh_host.hpp"
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <set>

static int fails = 0;
#define CHECK(c) do { if (!(c)) { if (fails++ < 10) fprintf(stderr, "FAILED %s (line %d)\n", #c, __LINE__); } } while (0)

// fills the w words with cross(e2, e1), as the device's records have them
static std::vector<float> with_normals(uint32_t n, std::vector<float> r)
{
    for (uint32_t i = 0; i < n; i++) {
        float *t = &r[12 * (size_t)i];
        const float *e1 = t + 4, *e2 = t + 8;
        t[3] = e2[1] * e1[2] - e2[2] * e1[1]; t[7] = e2[2] * e1[0] - e2[0] * e1[2]; t[11] = e2[0] * e1[1] - e2[1] * e1[0];
    }
    return r;
}

static void check(uint32_t n, const std::vector<float> &rec_in, uint32_t block, int threads, rt::HostTopology &T)
{
    const std::vector<float> rec = with_normals(n, rec_in);
    rt::sah_topology(n, rec.data(), block, threads, T);
    // orientation classes are parted before anything else: every class is one run of the leaf order
    {
        float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
        for (uint32_t i = 0; i < n; i++) for (int k = 0; k < 3; k++) {
            const float *r = &rec[12 * (size_t)i];
            for (float v : {r[k], r[k] + r[4 + k], r[k] + r[8 + k]}) { lo[k] = std::min(lo[k], v); hi[k] = std::max(hi[k], v); }
        }
        const float emax = std::max(hi[0] - lo[0], std::max(hi[1] - lo[1], hi[2] - lo[2]));
        const float thr = emax > 0.f ? 256.0f / (emax * emax) : 3.0e38f;
        std::vector<uint32_t> mask(rt::sah_detail::CELL_WORDS);
        rt::sah_detail::popular_cells(n, rec.data() + 4, rec.data() + 8, 12, mask.data());
        std::set<uint32_t> seen; uint32_t last = 0xffffffffu;
        for (uint32_t k = 0; k < n; k++) {
            const uint32_t cls = rt::sah_detail::normal_class(&rec[12 * (size_t)(T.keys[k] & 0xffffffffu)], thr, mask.data());
            if (cls != last) { CHECK(!seen.count(cls)); seen.insert(cls); last = cls; }
        }
    }
    CHECK(T.keys.size() == n && T.child.size() == 2 * (size_t)(n - 1) && T.range.size() == 2 * (size_t)(n - 1) && T.parent.size() == 2 * (size_t)n - 1);
    std::vector<char> seen(n, 0);
    for (uint32_t k = 0; k < n; k++) {
        CHECK((T.keys[k] >> 32) == k);
        const uint32_t t = (uint32_t)(T.keys[k] & 0xffffffffu);
        CHECK(t < n && !seen[t]);
        if (t < n) seen[t] = 1;
    }
    CHECK(T.parent[0] == 0xffffffffu);
    CHECK(T.range[0] == 0 && T.range[1] == n - 1);
    std::vector<char> referenced(2 * (size_t)n - 1, 0);
    std::set<uint32_t> top(T.top_list.begin(), T.top_list.end());
    CHECK(top.size() == T.top_list.size());
    size_t crossing = 0;
    for (uint32_t i = 0; i + 1 < n; i++) {
        const uint32_t first = T.range[2 * (size_t)i], last = T.range[2 * (size_t)i + 1];
        CHECK(first < last && last < n);
        uint32_t at = first;
        for (int side = 0; side < 2; side++) {
            const uint32_t c = T.child[2 * (size_t)i + side];
            CHECK(c < 2 * n - 1 && c != 0 && !referenced[c]);
            if (c >= 2 * n - 1) continue;
            referenced[c] = 1;
            CHECK(T.parent[c] == i);
            if (c >= n - 1) { CHECK(c - (n - 1) == at); at += 1; }
            else { CHECK(T.range[2 * (size_t)c] == at); at = T.range[2 * (size_t)c + 1] + 1; }
        }
        CHECK(at == last + 1);
        const bool cross = first / block != last / block;
        crossing += cross;
        CHECK(cross == (top.count(i) == 1));
    }
    CHECK(crossing == top.size());
    for (size_t c = 1; c < 2 * (size_t)n - 1; c++) CHECK(referenced[c]);
}

static std::vector<float> soup(uint32_t n, unsigned seed, float spread)
{
    std::mt19937 g(seed);
    std::uniform_real_distribution<float> pos(-10.f, 10.f), edge(-spread, spread);
    std::vector<float> r(12 * (size_t)n, 0.f);
    for (uint32_t i = 0; i < n; i++)
        for (int k = 0; k < 3; k++) { r[12 * (size_t)i + k] = pos(g); r[12 * (size_t)i + 4 + k] = edge(g); r[12 * (size_t)i + 8 + k] = edge(g); }
    return r;
}

static std::vector<float> grid(uint32_t cells)          // cells x cells quads of two triangles over a bumpy height
{
    std::vector<float> r;
    for (uint32_t y = 0; y < cells; y++) for (uint32_t x = 0; x < cells; x++) for (int half = 0; half < 2; half++) {
        auto hgt = [](float a, float b) { return 0.3f * std::sin(0.7f * a) * std::cos(0.5f * b); };
        float p[3][3];
        const float x0 = 0.2f * x, y0 = 0.2f * y, x1 = x0 + 0.2f, y1 = y0 + 0.2f;
        const float q[4][2] = {{x0, y0}, {x1, y0}, {x1, y1}, {x0, y1}};
        const int pick[2][3] = {{0, 1, 2}, {0, 2, 3}};
        for (int v = 0; v < 3; v++) { p[v][0] = q[pick[half][v]][0]; p[v][2] = q[pick[half][v]][1]; p[v][1] = hgt(p[v][0], p[v][2]); }
        for (int k = 0; k < 3; k++) r.push_back(p[0][k]);
        r.push_back(0.f);
        for (int k = 0; k < 3; k++) r.push_back(p[1][k] - p[0][k]);
        r.push_back(0.f);
        for (int k = 0; k < 3; k++) r.push_back(p[2][k] - p[0][k]);
        r.push_back(0.f);
    }
    return r;
}

int main(int argc, char **argv)
{
    rt::HostTopology A, B;
    for (uint32_t n : {2u, 3u, 4u, 5u, 7u, 64u, 257u, 1000u, 20000u, 70000u}) {
        const std::vector<float> r = soup(n, 100 + n, n % 2 ? 0.01f : 3.0f);
        check(n, r, 256, 1, A);
        check(n, r, 256, 5, B);
        CHECK(A.keys == B.keys && A.child == B.child && A.parent == B.parent && A.range == B.range);
        std::set<uint32_t> ta(A.top_list.begin(), A.top_list.end()), tb(B.top_list.begin(), B.top_list.end());
        CHECK(ta == tb);
    }
    {   // every triangle the same: no plane separates them, the ranges are halved
        std::vector<float> r = soup(1, 7, 1.0f), all;
        for (int i = 0; i < 1000; i++) all.insert(all.end(), r.begin(), r.end());
        check(1000, all, 256, 1, A);
    }
    {
        const std::vector<float> r = grid(100);
        check(20000, r, 256, 3, A);
    }
    if (argc > 1) {
        const uint32_t cells = (uint32_t)std::sqrt(atof(argv[1]) / 2.0);
        const std::vector<float> r = grid(cells);
        for (int threads : {1, 4, 8}) {
            const auto t0 = std::chrono::steady_clock::now();
            rt::sah_topology(2 * cells * cells, r.data(), 256, threads, A);
            printf("%u triangles, %d threads: %.2f ms\n", 2 * cells * cells, threads, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        }
    }
    if (fails) { fprintf(stderr, "%d checks failed\n", fails); return 1; }
    printf("ok\n");
    return 0;
}
