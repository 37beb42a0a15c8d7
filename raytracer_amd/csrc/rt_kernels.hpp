// rt_kernels.hpp — HIP kernels of the tracing path for gfx950 (wave64).
//
// Work decomposition (one frame, or one batch of user rays):
//   k_primary      closest hit of the camera rays; one thread per pixel, one wave per 8x8
//                  pixel tile, 4 sub-samples per thread in HIPEXEC_MODE_CPU
//   k_closest      closest hit of queued rays (bounces, user rays)
//   k_shadowgen    facing test per (hit, lamp); shadow rays appended to a queue with
//                  ballot/popcount compaction
//   k_anyhit       occlusion of queued shadow rays, early exit per wave
//   k_shade        texture fetch + Phong shading in the reference's accumulation order,
//                  bounce rays appended to the next level's queue (ballot/popcount)
//   k_resolve      folds the per-level colours back to front, averages sub-samples, tone
//                  curve, float4 RGBA store
// All scans are brute force over every triangle like the reference; triangle records are
// wave-uniform and reach the SIMDs through the scalar data path.
#pragma once
#include "rt_device.hpp"

namespace rt {

// per-call geometry of the pixel grid rendered by this call
struct Frame {
    int W, H;              // full frame
    int tile_first, tile_step, n_tiles;
    int nbx;               // 8x8 blocks per tile row = ceil(W/8)
    uint32_t n_slots;      // n_tiles * nbx * 64 pixel slots (one thread each)
    int spp;
};

struct RayRec { float4 a, b; };   // a = {o.xyz, d.x}  b = {d.y, d.z, sample index bits, lamp index bits}

DEV void load_ray(const float4 *rays, uint32_t j, f3 &o, f3 &d, uint32_t &s, uint32_t &aux) {
    float4 a = rays[2 * (size_t)j], b = rays[2 * (size_t)j + 1];
    o = mk(a.x, a.y, a.z); d = mk(a.w, b.x, b.y);
    s = __float_as_uint(b.z); aux = __float_as_uint(b.w);
}
DEV void store_ray(float4 *rays, uint32_t j, f3 o, f3 d, uint32_t s, uint32_t aux) {
    rays[2 * (size_t)j] = make_float4(o.x, o.y, o.z, d.x);
    rays[2 * (size_t)j + 1] = make_float4(d.y, d.z, __uint_as_float(s), __uint_as_float(aux));
}

// ---------------------------------------------------------------------------------------
// scene set-up: per-triangle records (runs once per hipexec_create)
// ---------------------------------------------------------------------------------------
constexpr float FILTER_KAPPA = 4e-6f;   // slack factor of the FILTER tests, ~5x the rounding bound

__global__ void k_setup_triangles(uint32_t n_tri, uint32_t n_pad, const float *tp, const float *te1, const float *te2,
                                  f3 cam, f3 center, float4 *tri_exact, float4 *tri_primx, float4 *tri_fastp,
                                  float4 *tri_fastg)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pad) return;
    float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i >= n_tri) {   // padding: det == 0 for every ray, FILTER slack < 0 rejects everything
        tri_exact[3 * i] = z; tri_exact[3 * i + 1] = z; tri_exact[3 * i + 2] = z;
        tri_primx[3 * i] = z; tri_primx[3 * i + 1] = z; tri_primx[3 * i + 2] = z;
        tri_fastp[3 * i] = make_float4(0.f, 0.f, 0.f, -1.f); tri_fastp[3 * i + 1] = z; tri_fastp[3 * i + 2] = z;
        for (int k = 0; k < 4; k++) tri_fastg[5 * i + k] = z;
        tri_fastg[5 * i + 4] = make_float4(0.f, 0.f, -1.f, -1.f);
        return;
    }
    f3 p = mk(tp[3 * i], tp[3 * i + 1], tp[3 * i + 2]);
    f3 e1 = mk(te1[3 * i], te1[3 * i + 1], te1[3 * i + 2]);
    f3 e2 = mk(te2[3 * i], te2[3 * i + 1], te2[3 * i + 2]);
    // EXACT: same operations as the per-ray code of the reference
    f3 c21 = cross(e2, e1);
    tri_exact[3 * i] = make_float4(p.x, p.y, p.z, c21.x);
    tri_exact[3 * i + 1] = make_float4(e1.x, e1.y, e1.z, c21.y);
    tri_exact[3 * i + 2] = make_float4(e2.x, e2.y, e2.z, c21.z);
    f3 q = cam - p;
    float dett = dot(q, cross(e1, e2));
    f3 ceq = cross(e2, q);
    f3 cqe = cross(q, e1);
    tri_primx[3 * i] = make_float4(c21.x, c21.y, c21.z, dett);
    tri_primx[3 * i + 1] = make_float4(ceq.x, ceq.y, ceq.z, 0.f);
    tri_primx[3 * i + 2] = make_float4(cqe.x, cqe.y, cqe.z, 0.f);
    // FILTER for camera rays: a hit needs t = dett/det > 0, i.e. sign(det) = sign(dett) = s, so
    // with everything multiplied by s the three edge functions must be >= 0 (up to slack).
    float nc = sqrtf(dot(c21, c21)), nu = sqrtf(dot(ceq, ceq)), nv = sqrtf(dot(cqe, cqe));
    float s = dett > 0.f ? 1.f : (dett < 0.f ? -1.f : 0.f);
    float slack = FILTER_KAPPA * (nc + nu + nv);
    if (s == 0.f || !(slack < 3.0e38f)) {
        // dett == 0: t == 0 can never pass.  Non-finite data: let the exact test decide.
        bool reject = (s == 0.f) && (slack < 3.0e38f);
        tri_fastp[3 * i] = make_float4(0.f, 0.f, 0.f, reject ? -1.f : 3.0e38f);
        tri_fastp[3 * i + 1] = z; tri_fastp[3 * i + 2] = z;
    } else {
        f3 cw = (c21 - ceq) - cqe;
        tri_fastp[3 * i] = make_float4(s * ceq.x, s * ceq.y, s * ceq.z, slack);
        tri_fastp[3 * i + 1] = make_float4(s * cqe.x, s * cqe.y, s * cqe.z, 0.f);
        tri_fastp[3 * i + 2] = make_float4(s * cw.x, s * cw.y, s * cw.z, 0.f);
    }
    // FILTER for arbitrary rays, in coordinates translated by -center.  With o' = o - center,
    // mo = cross(o', d), p' = p - center, k1 = cross(p', e1), k2 = cross(e2, p'):
    //   detu = dot(e2, mo) - dot(d, k2)
    //   detv = -dot(e1, mo) - dot(d, k1)
    //   w    = det - detu - detv = dot(d, c21 + k2 + k1) + dot(e1 - e2, mo)
    f3 pp = p - center;
    f3 k1 = cross(pp, e1), k2 = cross(e2, pp);
    f3 cwd = (c21 + k2) + k1, cwm = e1 - e2;
    float ne = sqrtf(dot(e1, e1)) + sqrtf(dot(e2, e2));
    float A = FILTER_KAPPA * ne;
    float B = FILTER_KAPPA * (ne * sqrtf(dot(pp, pp)) + nc);
    if (!(A < 3.0e38f) || !(B < 3.0e38f)) { A = 3.0e38f; B = 3.0e38f; }
    tri_fastg[5 * i] = make_float4(-k2.x, -k2.y, -k2.z, e2.x);
    tri_fastg[5 * i + 1] = make_float4(e2.y, e2.z, -k1.x, -k1.y);
    tri_fastg[5 * i + 2] = make_float4(-k1.z, -e1.x, -e1.y, -e1.z);
    tri_fastg[5 * i + 3] = make_float4(cwd.x, cwd.y, cwd.z, cwm.x);
    tri_fastg[5 * i + 4] = make_float4(cwm.y, cwm.z, A, B);
}

// ---------------------------------------------------------------------------------------
// FILTER tests
// ---------------------------------------------------------------------------------------
DEV float dot3_fma(float ax, float ay, float az, f3 b) { return fmaf(az, b.z, fmaf(ay, b.y, ax * b.x)); }

// camera ray vs FILTER camera record; true = may be a hit
DEV bool filter_cam(f3 d, float4 a, float4 b, float4 c) {
    float su = dot3_fma(a.x, a.y, a.z, d);
    float sv = dot3_fma(b.x, b.y, b.z, d);
    float sw = dot3_fma(c.x, c.y, c.z, d);
    float m = fminf(fminf(su, sv), sw);
    return !(m < -a.w);
}

// per-ray constants of the general FILTER
struct RayF { f3 d, mo; float ro, rd; };
DEV RayF rayf_make(const DScene &S, f3 o, f3 d) {
    RayF r;
    f3 op = mk(o.x - S.center[0], o.y - S.center[1], o.z - S.center[2]);
    r.d = d;
    r.mo = cross(op, d);
    r.rd = sqrtf(dot(d, d)) * 1.0001f;
    r.ro = sqrtf(dot(op, op)) * r.rd;
    return r;
}
// arbitrary ray vs general FILTER record (5 float4); true = may be a hit
DEV bool filter_gen(const RayF &r, float4 f0, float4 f1, float4 f2, float4 f3_, float4 f4) {
    float su = fmaf(f1.y, r.mo.z, fmaf(f1.x, r.mo.y, fmaf(f0.w, r.mo.x, dot3_fma(f0.x, f0.y, f0.z, r.d))));
    float sv = fmaf(f2.w, r.mo.z, fmaf(f2.z, r.mo.y, fmaf(f2.y, r.mo.x, dot3_fma(f1.z, f1.w, f2.x, r.d))));
    float sw = fmaf(f4.y, r.mo.z, fmaf(f4.x, r.mo.y, fmaf(f3_.w, r.mo.x, dot3_fma(f3_.x, f3_.y, f3_.z, r.d))));
    float slack = fmaf(f4.z, r.ro, f4.w * r.rd);
    float mn = fminf(fminf(su, sv), sw);
    float mx = fmaxf(fmaxf(su, sv), sw);
    return !(mn < -slack && mx > slack);
}

// ---------------------------------------------------------------------------------------
// k_primary
// ---------------------------------------------------------------------------------------
DEV bool slot_to_pixel(const Frame &F, uint32_t slot, int &x, int &y, int &lrow) {
    uint32_t blk = slot >> 6, lane = slot & 63;
    int k = (int)(blk / (uint32_t)F.nbx), bx = (int)(blk % (uint32_t)F.nbx);
    x = bx * 8 + (int)(lane & 7);
    int ly = (int)(lane >> 3);
    lrow = k * HIPEXEC_TILE_ROWS + ly;
    y = (F.tile_first + k * F.tile_step) * HIPEXEC_TILE_ROWS + ly;
    return slot < F.n_slots && x < F.W && y < F.H;
}

// primary ray directions of pixel (x, y): ray_tracer.cpp:25-45 (4 sub-samples) /
// ray_tracer_cl.cpp:69-85 (1 sample)
template <int MODE>
DEV void camera_rays(const DScene &S, const Frame &F, int x, int y, f3 *d) {
    float camHeight = 0.5f;
    float camWidth = (float)F.W * (camHeight / (float)F.H);
    float camDist = 1.0f;
    float dh = camHeight / (float)F.H;
    float dw = camWidth / (float)F.W;
    float rayX = -(camWidth / 2) + (float)x * dw;
    float rayY = -(camHeight / 2) + (float)y * dh;
    f3 world = vecmat(mk(rayX, rayY, -camDist), S.cam_mat);
    if (MODE == HIPEXEC_MODE_CL) {
        d[0] = normalize(world);
    } else {
        f3 wdx = vecmat(mk(dw / 2.0f, 0.0f, 0.0f), S.cam_mat);
        f3 wdy = vecmat(mk(0.0f, dh / 2.0f, 0.0f), S.cam_mat);
        d[0] = normalize(world);
        d[1] = normalize(world + wdx);
        d[2] = normalize(world + wdy);
        d[3] = normalize((world + wdx) + wdy);
    }
}

template <int MODE, bool FILTER>
__global__ __launch_bounds__(256) void k_primary(DScene S, Frame F, float4 *__restrict__ rays, float4 *__restrict__ hits)
{
    constexpr int SPP = (MODE == HIPEXEC_MODE_CL) ? 1 : 4;
    constexpr int U = 4;
    uint32_t slot = blockIdx.x * 256u + threadIdx.x;
    int x, y, lrow;
    bool active = slot_to_pixel(F, slot, x, y, lrow);
    if (!active) { x = min(x, F.W - 1); y = min(y, F.H - 1); }     // keep idle lanes on finite data
    f3 d[SPP];
    camera_rays<MODE>(S, F, x, y, d);
    Best tb[SPP];
#pragma unroll
    for (int r = 0; r < SPP; r++) tb[r] = best_none();

    const float4 *__restrict__ FP = S.tri_fastp;
    const float4 *__restrict__ PX = S.tri_primx;
    for (uint32_t i = 0; i < S.n_tri_pad; i += U) {
        bool any = !FILTER;
        if (FILTER) {
#pragma unroll
            for (int k = 0; k < U; k++) {
                float4 a = FP[3 * (i + k)], b = FP[3 * (i + k) + 1], c = FP[3 * (i + k) + 2];
#pragma unroll
                for (int r = 0; r < SPP; r++) any |= filter_cam(d[r], a, b, c);
            }
        }
        if (__ballot(any)) {
#pragma unroll 1
            for (int k = 0; k < U; k++) {
                uint32_t ti = i + k;
                float4 a = FP[3 * ti], b = FP[3 * ti + 1], c = FP[3 * ti + 2];
                float4 e0 = PX[3 * ti], e1 = PX[3 * ti + 1], e2 = PX[3 * ti + 2];
#pragma unroll
                for (int r = 0; r < SPP; r++) {
                    if (!FILTER || filter_cam(d[r], a, b, c)) {
                        float t, u, v;
                        if (tri_exact_cam<MODE>(d[r], xyz(e0), e0.w, xyz(e1), xyz(e2), t, u, v))
                            best_update<MODE>(tb[r], t, u, v, (int)ti);
                    }
                }
            }
        }
    }
    f3 o = mk(S.cam_pos[0], S.cam_pos[1], S.cam_pos[2]);
#pragma unroll
    for (int r = 0; r < SPP; r++) {
        Best sb = spheres_closest<MODE>(S, o, d[r]);
        float4 h = merge_hits<MODE>(S, tb[r], sb);
        if (!active) h.w = __int_as_float(-2);            // pixel slot outside the frame
        uint32_t s = (uint32_t)r * F.n_slots + slot;
        if (slot < F.n_slots) {
            hits[s] = h;
            store_ray(rays, s, o, d[r], s, 0u);
        }
    }
}

// ---------------------------------------------------------------------------------------
// k_closest: closest hit of n queued rays, R rays per lane
// ---------------------------------------------------------------------------------------
template <int MODE, int R, bool FILTER>
__global__ __launch_bounds__(256) void k_closest(DScene S, const float4 *__restrict__ rays, uint32_t n,
                                                 float4 *__restrict__ hits)
{
    constexpr int U = 4;
    uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6;
    uint32_t lane = threadIdx.x & 63u;
    uint32_t base = wave * 64u * R + lane;
    f3 o[R], d[R]; RayF rf[R]; Best tb[R]; bool valid[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        uint32_t j = base + 64u * r;
        valid[r] = j < n;
        uint32_t s, aux;
        load_ray(rays, valid[r] ? j : 0u, o[r], d[r], s, aux);
        rf[r] = rayf_make(S, o[r], d[r]);
        tb[r] = best_none();
    }
    if (n == 0) return;
    const float4 *__restrict__ FG = S.tri_fastg;
    const float4 *__restrict__ EX = S.tri_exact;
    for (uint32_t i = 0; i < S.n_tri_pad; i += U) {
        bool any = !FILTER;
        if (FILTER) {
#pragma unroll
            for (int k = 0; k < U; k++) {
                const float4 *g = FG + 5 * (size_t)(i + k);
                float4 f0 = g[0], f1 = g[1], f2 = g[2], f3_ = g[3], f4 = g[4];
#pragma unroll
                for (int r = 0; r < R; r++) any |= filter_gen(rf[r], f0, f1, f2, f3_, f4);
            }
        }
        if (__ballot(any)) {
#pragma unroll 1
            for (int k = 0; k < U; k++) {
                uint32_t ti = i + k;
                const float4 *g = FG + 5 * (size_t)ti;
                float4 f0 = g[0], f1 = g[1], f2 = g[2], f3_ = g[3], f4 = g[4];
                float4 e0 = EX[3 * ti], e1 = EX[3 * ti + 1], e2 = EX[3 * ti + 2];
#pragma unroll
                for (int r = 0; r < R; r++) {
                    if (!FILTER || filter_gen(rf[r], f0, f1, f2, f3_, f4)) {
                        float t, u, v;
                        if (ti < S.n_tri && tri_exact<MODE>(o[r], d[r], xyz(e0), xyz(e1), xyz(e2), t, u, v))
                            best_update<MODE>(tb[r], t, u, v, (int)ti);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
        Best sb = spheres_closest<MODE>(S, o[r], d[r]);
        float4 h = merge_hits<MODE>(S, tb[r], sb);
        if (valid[r]) hits[base + 64u * r] = h;
    }
}

// ---------------------------------------------------------------------------------------
// k_anyhit: occlusion of n queued shadow rays.  Sets bit `lamp` of occl[s * lw + lamp/32]
// for occluded rays and accumulates the algorithmic test counts (reference scan order:
// triangles by index, then spheres; a scan stops at its first hit).
// counters[0] += triangle tests, counters[1] += sphere tests.
// ---------------------------------------------------------------------------------------
template <int MODE, bool FILTER>
__global__ __launch_bounds__(256) void k_anyhit(DScene S, const float4 *__restrict__ rays, uint32_t n,
                                                uint32_t *__restrict__ occl, uint32_t lw,
                                                unsigned long long *__restrict__ counters)
{
    constexpr int U = 4;
    uint32_t j = blockIdx.x * 256u + threadIdx.x;
    bool valid = j < n;
    f3 o, d; uint32_t s, lamp;
    load_ray(rays, valid ? j : 0u, o, d, s, lamp);
    RayF rf = rayf_make(S, o, d);
    bool done = !valid;
    uint32_t tri_tests = valid ? S.n_tri : 0u, sph_tests = valid ? S.n_sph : 0u;
    bool occluded = false;
    const float4 *__restrict__ FG = S.tri_fastg;
    const float4 *__restrict__ EX = S.tri_exact;
    for (uint32_t i = 0; i < S.n_tri_pad; i += U) {
        if (__ballot(!done) == 0) break;
        bool any = !FILTER && !done;
        if (FILTER) {
#pragma unroll
            for (int k = 0; k < U; k++) {
                const float4 *g = FG + 5 * (size_t)(i + k);
                any |= filter_gen(rf, g[0], g[1], g[2], g[3], g[4]);
            }
            any = any && !done;
        }
        if (__ballot(any)) {
#pragma unroll 1
            for (int k = 0; k < U; k++) {
                uint32_t ti = i + k;
                const float4 *g = FG + 5 * (size_t)ti;
                float4 e0 = EX[3 * ti], e1 = EX[3 * ti + 1], e2 = EX[3 * ti + 2];
                if (!done && ti < S.n_tri && (!FILTER || filter_gen(rf, g[0], g[1], g[2], g[3], g[4]))) {
                    float t, u, v;
                    if (tri_exact<MODE>(o, d, xyz(e0), xyz(e1), xyz(e2), t, u, v)) {
                        done = true; occluded = true; tri_tests = ti + 1; sph_tests = 0;
                    }
                }
            }
        }
    }
    if (!done) {
        for (uint32_t k = 0; k < S.n_sph; k++) {
            float4 sp = S.sph[k];
            float t;
            if (!done && sph_exact<MODE>(o, d, xyz(sp), sp.w, t)) { done = true; occluded = true; sph_tests = k + 1; }
        }
    }
    if (occluded) atomicOr(&occl[(size_t)s * lw + (lamp >> 5)], 1u << (lamp & 31u));
    unsigned long long a = wave_sum_u64(tri_tests), b = wave_sum_u64(sph_tests);
    if ((threadIdx.x & 63u) == 0) {
        if (a) atomicAdd(&counters[0], a);
        if (b) atomicAdd(&counters[1], b);
    }
}

// ---------------------------------------------------------------------------------------
// k_shadowgen: one thread per ray of the current level
// ---------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void k_shadowgen(DScene S, const float4 *__restrict__ rays, const float4 *__restrict__ hits,
                                                   uint32_t n, float4 *__restrict__ shadow_q, uint32_t *__restrict__ q_count)
{
    uint32_t j = blockIdx.x * 256u + threadIdx.x;
    bool valid = j < n;
    f3 o, d; uint32_t s, aux;
    load_ray(rays, valid ? j : 0u, o, d, s, aux);
    float4 h = hits[valid ? j : 0u];
    bool hit = valid && __float_as_int(h.w) >= 0;
    Surf sf;
    if (hit) sf = surface(S, o, d, h, false);
    else { sf.P = mk(0.f, 0.f, 0.f); sf.N = mk(0.f, 0.f, 1.f); }
    for (uint32_t l = 0; l < S.n_lamp; l++) {
        float4 lp = S.lamp[l];
        f3 toLamp = xyz(lp) - sf.P;
        float dotWithLamp = dot(toLamp, sf.N);
        float dotWithDir = dot(d, sf.N);
        bool cast = hit && casts_shadow_ray<MODE>(dotWithDir, dotWithLamp);
        uint32_t pos = wave_append(cast, q_count);
        if (cast) store_ray(shadow_q, pos, sf.P, toLamp, s, l);
    }
}

// ---------------------------------------------------------------------------------------
// k_shade: one thread per ray of the current level.  Writes local[s] = {colour, w} where
// w = reflectionFactor if a bounce ray was queued and -1 otherwise, clears the occlusion bits
// of s, and appends the bounce ray (ray_tracer.cpp:190-196) to next_rays.
// ---------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void k_shade(DScene S, const float4 *__restrict__ rays, const float4 *__restrict__ hits,
                                               uint32_t n, uint32_t *__restrict__ occl, uint32_t lw,
                                               float4 *__restrict__ local, bool spawn,
                                               float4 *__restrict__ next_rays, uint32_t *__restrict__ next_count)
{
    uint32_t j = blockIdx.x * 256u + threadIdx.x;
    bool valid = j < n;
    f3 o, d; uint32_t s, aux;
    load_ray(rays, valid ? j : 0u, o, d, s, aux);
    float4 h = hits[valid ? j : 0u];
    int id = __float_as_int(h.w);
    bool bounce = false;
    f3 bo = mk(0.f, 0.f, 0.f), bd = mk(0.f, 0.f, 1.f);
    if (valid && id >= -1) {
        f3 ret; float w = -1.0f;
        if (id < 0) {
            // miss: ray_tracer.cpp:199 returns the horizon; ray_tracer_cl.cpp:119-123 adds it to 0
            f3 hz = mk(S.horizon[0], S.horizon[1], S.horizon[2]);
            ret = (MODE == HIPEXEC_MODE_CL) ? (mk(0.f, 0.f, 0.f) + hz) : hz;
        } else {
            Surf sf = surface(S, o, d, h, true);
            ret = (MODE == HIPEXEC_MODE_CL) ? mk(0.f, 0.f, 0.f) : mk(S.ambient[0], S.ambient[1], S.ambient[2]);
            for (uint32_t l = 0; l < S.n_lamp; l++) {
                float4 lp = S.lamp[l];
                f3 toLamp = xyz(lp) - sf.P;
                float dotWithLamp = dot(toLamp, sf.N);
                float dotWithDir = dot(d, sf.N);
                if (!casts_shadow_ray<MODE>(dotWithDir, dotWithLamp)) continue;
                if (occl[(size_t)s * lw + (l >> 5)] & (1u << (l & 31u))) continue;
                ret = add_lamp<MODE>(ret, sf, d, toLamp, lp.w, S.lamp_dist[l]);
            }
            if (MODE == HIPEXEC_MODE_CPU && spawn && sf.kr > RT_EPSF) {
                bounce = true; w = sf.kr;
                bo = sf.P;
                bd = d - 2.0f * sf.N * dot(d, sf.N);
            }
        }
        local[s] = make_float4(ret.x, ret.y, ret.z, w);
        for (uint32_t k = 0; k < lw; k++) occl[(size_t)s * lw + k] = 0u;
    }
    if (spawn) {
        uint32_t pos = wave_append(bounce, next_count);
        if (bounce) store_ray(next_rays, pos, bo, bd, s, 0u);
    }
}

// ---------------------------------------------------------------------------------------
// k_resolve: fold levels (ret += traced * kr, ray_tracer.cpp:192-194), average the 4
// sub-samples and apply the tone curve (ray_tracer.cpp:42-51 / ray_tracer_cl.cpp:97-101)
// ---------------------------------------------------------------------------------------
DEV f3 fold_levels(const float4 *__restrict__ local, size_t level_stride, int n_levels, uint32_t s) {
    int m = 0;
    while (m + 1 < n_levels && local[(size_t)m * level_stride + s].w > 0.0f) m++;
    float4 c = local[(size_t)m * level_stride + s];
    f3 total = mk(c.x, c.y, c.z);
    for (int k = m - 1; k >= 0; k--) {
        float4 lk = local[(size_t)k * level_stride + s];
        total = mk(lk.x, lk.y, lk.z) + total * lk.w;
    }
    return total;
}

template <int MODE>
__global__ __launch_bounds__(256) void k_resolve(Frame F, const float4 *__restrict__ local, size_t level_stride, int n_levels,
                                                 const float4 *__restrict__ hits0, int tone,
                                                 float4 *__restrict__ out_rgba, int32_t *__restrict__ out_ids)
{
    constexpr int SPP = (MODE == HIPEXEC_MODE_CL) ? 1 : 4;
    uint32_t slot = blockIdx.x * 256u + threadIdx.x;
    int x, y, lrow;
    if (!slot_to_pixel(F, slot, x, y, lrow)) return;
    f3 c = mk(0.f, 0.f, 0.f);
#pragma unroll
    for (int r = 0; r < SPP; r++) {
        uint32_t s = (uint32_t)r * F.n_slots + slot;
        f3 t = fold_levels(local, level_stride, n_levels, s);
        c = (r == 0) ? t : (c + t);
        if (out_ids) out_ids[((size_t)lrow * F.W + x) * SPP + r] = __float_as_int(hits0[s].w);
    }
    if (MODE == HIPEXEC_MODE_CPU) c = c / 4.0f;
    if (tone) c = mk(pow_cr(c.x / 2.2f, 0.3f), pow_cr(c.y / 2.2f, 0.3f), pow_cr(c.z / 2.2f, 0.3f));
    out_rgba[(size_t)lrow * F.W + x] = make_float4(c.x, c.y, c.z, 1.0f);
}

// batch variant for hipexec_trace_rays: colours3[i] += traced colour
__global__ void k_resolve_batch(uint32_t n, const float4 *__restrict__ local, size_t level_stride, int n_levels,
                                float *__restrict__ colours3)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    f3 t = fold_levels(local, level_stride, n_levels, i);
    colours3[3 * (size_t)i] = colours3[3 * (size_t)i] + t.x;
    colours3[3 * (size_t)i + 1] = colours3[3 * (size_t)i + 1] + t.y;
    colours3[3 * (size_t)i + 2] = colours3[3 * (size_t)i + 2] + t.z;
}

// float4 RGBA tiles -> float RGB rows of the full frame (image_bitmap layout)
__global__ void k_pack_rgb(Frame F, const float4 *__restrict__ rgba, float *__restrict__ rgb, const int32_t *__restrict__ ids_in,
                           int32_t *__restrict__ ids_out, int spp)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t total = (uint32_t)F.n_tiles * HIPEXEC_TILE_ROWS * (uint32_t)F.W;
    if (i >= total) return;
    int lrow = (int)(i / (uint32_t)F.W), x = (int)(i % (uint32_t)F.W);
    int k = lrow / HIPEXEC_TILE_ROWS, ly = lrow % HIPEXEC_TILE_ROWS;
    int y = (F.tile_first + k * F.tile_step) * HIPEXEC_TILE_ROWS + ly;
    if (y >= F.H) return;
    float4 c = rgba[i];
    size_t p = (size_t)y * F.W + x;
    rgb[3 * p] = c.x; rgb[3 * p + 1] = c.y; rgb[3 * p + 2] = c.z;
    if (ids_out) for (int r = 0; r < spp; r++) ids_out[p * spp + r] = ids_in[(size_t)i * spp + r];
}

// full hit records for hipexec_closest_hits (the `Hit` of scene.h:48-63)
__global__ void k_export_hits(DScene S, const float4 *__restrict__ rays, const float4 *__restrict__ hits, uint32_t n,
                              hipexec_hit *__restrict__ out)
{
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    f3 o, d; uint32_t s, aux;
    load_ray(rays, j, o, d, s, aux);
    float4 h = hits[j];
    hipexec_hit r;
    r.id = __float_as_int(h.w); r.mat = -1; r.t = 0.f; r.u = 0.f; r.v = 0.f;
    for (int k = 0; k < 3; k++) { r.point[k] = 0.f; r.norm[k] = 0.f; r.color[k] = 0.f; }
    if (r.id >= 0) {
        Surf sf = surface(S, o, d, h, true);
        r.t = h.x; r.u = h.y; r.v = h.z;
        r.mat = (sf.mat < (int)S.n_mat) ? sf.mat : -1;
        r.point[0] = sf.P.x; r.point[1] = sf.P.y; r.point[2] = sf.P.z;
        r.norm[0] = sf.N.x; r.norm[1] = sf.N.y; r.norm[2] = sf.N.z;
        r.color[0] = sf.color.x; r.color[1] = sf.color.y; r.color[2] = sf.color.z;
    }
    out[j] = r;
}

}  // namespace rt
