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
// k_primary / k_closest / k_anyhit* are the index-order scan algorithm: triangles are visited in
// the reference's order (optionally skipping 128-triangle chunks no ray of the wave can reach),
// records are wave-uniform and reach the SIMDs through the scalar data path or LDS.  The default
// algorithm, the hierarchy, lives in rt_bvh.hpp and shares k_shade / k_resolve with this file.
#pragma once
#include "rt_device.hpp"

namespace rt {

// per-call geometry of the pixel grid rendered by this call
struct Frame {
    int W, H;              // full frame
    int tile_first, tile_step, n_tiles;
    int out_first, out_step;   // tile k of this call is stored as tile out_first + k * out_step of the output buffer
    int nbx;               // 8x8 blocks per tile row of this call
    uint32_t n_slots;      // n_tiles * nbx * 64 pixel slots (one thread each)
    int spp;
    // Window of the frame this call renders: pixels [cx0, cx1) x [cy0, cy1) (the whole frame, or the sub-rectangle of
    // a block-wise caller, ray_tracer_cl.cpp:59-105).  Block column 0 of the call starts at pixel x_blk0 * 8; the
    // output buffer holds out_w pixels per row, its column 0 being pixel cx0.
    int x_blk0, out_w, cx0, cx1, cy0, cy1;
};

struct RayRec { float4 a, b; };   // a = {o.xyz, d.x}  b = {d.y, d.z, sample index bits, lamp index bits}

DEV void load_ray(const float4 *rays, uint32_t j, f3 &o, f3 &d, uint32_t &s, uint32_t &aux) {
    float4 a = rays[2 * (size_t)j], b = rays[2 * (size_t)j + 1];
    o = mk(a.x, a.y, a.z); d = mk(a.w, b.x, b.y);
    s = __float_as_uint(b.z); aux = __float_as_uint(b.w);
}
// Fill of a ray queue: `n` is the size the host launched for (known exactly, or remembered from an identical earlier
// batch); n_dev, when given, is the device counter the queue's producers added to.  A kernel never reads past either.
DEV uint32_t queue_fill(uint32_t n, const uint32_t *__restrict__ n_dev) { return n_dev ? min(n, *n_dev) : n; }

DEV void store_ray(float4 *rays, uint32_t j, f3 o, f3 d, uint32_t s, uint32_t aux) {
    rays[2 * (size_t)j] = make_float4(o.x, o.y, o.z, d.x);
    rays[2 * (size_t)j + 1] = make_float4(d.y, d.z, __uint_as_float(s), __uint_as_float(aux));
}

// ---------------------------------------------------------------------------------------
// scene set-up: per-triangle records (runs once per hipexec_create)
// ---------------------------------------------------------------------------------------
// Triangle-pair records are staged through LDS in chunks of LDS_CHUNK pairs: the 4 waves of a
// workgroup fetch each chunk from L2/HBM once (coalesced 16-byte loads, double buffered, next
// chunk in flight during the tests of the current one) and read it back with uniform-address
// ds_read_b128 broadcasts.
constexpr int LDS_CHUNK = 64;

constexpr float FILTER_KAPPA = 4e-6f;   // slack factor of the FILTER tests, ~5x the rounding bound
constexpr float FILTER_NEVER = 1e30f;   // coefficient of a record that must never pass

DEV void set_half(v2f &dst, int h, float v) { if (h) dst.y = v; else dst.x = v; }
DEV void set_cam_half(PairCam &pc, int h, f3 cu, f3 cv, f3 cw) {
    set_half(pc.ux, h, cu.x); set_half(pc.uy, h, cu.y); set_half(pc.uz, h, cu.z);
    set_half(pc.vx, h, cv.x); set_half(pc.vy, h, cv.y); set_half(pc.vz, h, cv.z);
    set_half(pc.wx, h, cw.x); set_half(pc.wy, h, cw.y); set_half(pc.wz, h, cw.z);
    set_half(pc.pad, h, 0.f);
}

// EXACT records alone (what the hierarchy build and both algorithms' exact tests read): one thread per triangle slot.
__global__ void k_setup_exact(uint32_t n_tri, uint32_t n_pad, const float *tp, const float *te1, const float *te2, float4 *tri_exact)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pad) return;
    float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = r0;
    if (i < n_tri) {
        f3 p = mk(tp[3 * i], tp[3 * i + 1], tp[3 * i + 2]);
        f3 e1 = mk(te1[3 * i], te1[3 * i + 1], te1[3 * i + 2]);
        f3 e2 = mk(te2[3 * i], te2[3 * i + 1], te2[3 * i + 2]);
        f3 c21 = cross(e2, e1);                      // the same operations as in k_setup_cam_pairs
        r0 = make_float4(p.x, p.y, p.z, c21.x); r1 = make_float4(e1.x, e1.y, e1.z, c21.y); r2 = make_float4(e2.x, e2.y, e2.z, c21.z);
    }
    tri_exact[3 * (size_t)i] = r0; tri_exact[3 * (size_t)i + 1] = r1; tri_exact[3 * (size_t)i + 2] = r2;
}

// Camera-dependent records of the index-order scans, one thread per pair of triangles (2j, 2j+1): the EXACT
// camera-ray records (what depends only on camera and triangle of the exact test, from the same operations) and the
// FILTER pair records of the camera rays.  Made per executor and again when its camera changes.
__global__ void k_setup_cam_pairs(uint32_t n_tri, uint32_t n_pad, const float *tp, const float *te1, const float *te2,
                                  f3 cam, float4 *tri_primx, PairCam *pair_cam)
{
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * j >= n_pad) return;
    float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    PairCam pc;
    for (int h = 0; h < 2; h++) {
        uint32_t i = 2 * j + h;
        f3 zero = mk(0.f, 0.f, 0.f);
        // default: never passes (padding, zero-area triangles)
        f3 cu = mk(FILTER_NEVER, 0.f, 0.f), cv = mk(-FILTER_NEVER, 0.f, 0.f), cw = zero;
        if (i >= n_tri) {
            tri_primx[3 * i] = z; tri_primx[3 * i + 1] = z; tri_primx[3 * i + 2] = z;
        } else {
            f3 p = mk(tp[3 * i], tp[3 * i + 1], tp[3 * i + 2]);
            f3 e1 = mk(te1[3 * i], te1[3 * i + 1], te1[3 * i + 2]);
            f3 e2 = mk(te2[3 * i], te2[3 * i + 1], te2[3 * i + 2]);
            // EXACT: same operations as the per-ray code of the reference
            f3 c21 = cross(e2, e1);
            f3 q = cam - p;
            float dett = dot(q, cross(e1, e2));
            f3 ceq = cross(e2, q);
            f3 cqe = cross(q, e1);
            tri_primx[3 * i] = make_float4(c21.x, c21.y, c21.z, dett);
            tri_primx[3 * i + 1] = make_float4(ceq.x, ceq.y, ceq.z, 0.f);
            tri_primx[3 * i + 2] = make_float4(cqe.x, cqe.y, cqe.z, 0.f);
            float nc = sqrtf(dot(c21, c21)), nu = sqrtf(dot(ceq, ceq)), nv = sqrtf(dot(cqe, cqe));
            float slack_c = FILTER_KAPPA * (nc + nu + nv);
            if (!(slack_c < 3.0e38f)) {
                cu = zero; cv = zero;                 // let the exact test decide: all-zero records always pass
            } else if (nc > 0.f) {
                // camera rays: a hit needs t = dett/det > 0, i.e. sign(det) = sign(dett) = s; with
                // everything multiplied by s the three edge functions must be >= 0 (up to slack)
                float s = dett > 0.f ? 1.f : (dett < 0.f ? -1.f : 0.f);
                if (s != 0.f) {
                    float k = s / slack_c;
                    cu = ceq * k; cv = cqe * k; cw = ((c21 - ceq) - cqe) * k;
                }
            }
            // nc == 0: det == 0 for every ray, the exact tests always reject -> never passes
        }
        set_cam_half(pc, h, cu, cv, cw);
    }
    pair_cam[j] = pc;
}

// FILTER pair records for arbitrary rays, one thread per pair of triangles (2j, 2j+1).  `rmax` bounds the distance of
// ray origins from `center` the slack is valid for (farther origins scale their threshold).  Camera independent:
// shared by every executor of the scene.
__global__ void k_setup_gen_pairs(uint32_t n_tri, uint32_t n_pad, const float *tp, const float *te1, const float *te2,
                                  f3 center, float rmax, PairGen *pair_gen)
{
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * j >= n_pad) return;
    PairGen pg;
    for (int h = 0; h < 2; h++) {
        uint32_t i = 2 * j + h;
        f3 zero = mk(0.f, 0.f, 0.f);
        f3 ud = mk(FILTER_NEVER, 0.f, 0.f), um = zero, vd = mk(-FILTER_NEVER, 0.f, 0.f), vm = zero, wd = zero, wm = zero;
        if (i < n_tri) {
            f3 p = mk(tp[3 * i], tp[3 * i + 1], tp[3 * i + 2]);
            f3 e1 = mk(te1[3 * i], te1[3 * i + 1], te1[3 * i + 2]);
            f3 e2 = mk(te2[3 * i], te2[3 * i + 1], te2[3 * i + 2]);
            f3 c21 = cross(e2, e1);
            float nc = sqrtf(dot(c21, c21));
            f3 pp = p - center;
            float ne = sqrtf(dot(e1, e1)) + sqrtf(dot(e2, e2));
            float slack_g = FILTER_KAPPA * (ne * (rmax + sqrtf(dot(pp, pp))) + nc);
            if (!(slack_g < 3.0e38f)) {
                ud = zero; vd = zero;                 // let the exact test decide
            } else if (nc > 0.f) {
                // coordinates translated by -center: with o' = o - center, mo = cross(o', d), p' = p - center,
                // k1 = cross(p', e1), k2 = cross(e2, p'):
                //   detu = dot(e2, mo) - dot(d, k2)
                //   detv = -dot(e1, mo) - dot(d, k1)
                //   w    = det - detu - detv = dot(d, c21 + k2 + k1) + dot(e1 - e2, mo)
                f3 k1 = cross(pp, e1), k2 = cross(e2, pp);
                float g = 1.0f / slack_g;
                ud = (-k2) * g; um = e2 * g;
                vd = (-k1) * g; vm = (-e1) * g;
                wd = ((c21 + k2) + k1) * g; wm = (e1 - e2) * g;
            }
        }
        set_half(pg.ud[0], h, ud.x); set_half(pg.ud[1], h, ud.y); set_half(pg.ud[2], h, ud.z);
        set_half(pg.um[0], h, um.x); set_half(pg.um[1], h, um.y); set_half(pg.um[2], h, um.z);
        set_half(pg.vd[0], h, vd.x); set_half(pg.vd[1], h, vd.y); set_half(pg.vd[2], h, vd.z);
        set_half(pg.vm[0], h, vm.x); set_half(pg.vm[1], h, vm.y); set_half(pg.vm[2], h, vm.z);
        set_half(pg.wd[0], h, wd.x); set_half(pg.wd[1], h, wd.y); set_half(pg.wd[2], h, wd.z);
        set_half(pg.wm[0], h, wm.x); set_half(pg.wm[1], h, wm.y); set_half(pg.wm[2], h, wm.z);
    }
    pair_gen[j] = pg;
}

// FILTER records of the lines through one lamp position L: a shadow ray is (P, L - P), which
// lies on the line through L with direction d = L - P up to the rounding of that subtraction.
// With q = L - p the edge functions dot(d, cross(e2,q)), dot(d, cross(q,e1)) and
// w = dot(d, c21) - u - v have one sign on lines that pierce the triangle.  The slack covers
// the rounding of the chains and the offset between the two lines (<= 2^-23 * (|L| + |P|),
// |P| <= abs_bound for hit points of this scene).
__global__ void k_setup_lamp_pairs(uint32_t n_tri, uint32_t n_pad, const float *tp, const float *te1, const float *te2,
                                   f3 L, float abs_bound, PairCam *pair_lamp)
{
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * j >= n_pad) return;
    PairCam pc;
    for (int h = 0; h < 2; h++) {
        uint32_t i = 2 * j + h;
        f3 zero = mk(0.f, 0.f, 0.f);
        f3 cu = mk(FILTER_NEVER, 0.f, 0.f), cv = mk(-FILTER_NEVER, 0.f, 0.f), cw = zero;
        if (i < n_tri) {
            f3 p = mk(tp[3 * i], tp[3 * i + 1], tp[3 * i + 2]);
            f3 e1 = mk(te1[3 * i], te1[3 * i + 1], te1[3 * i + 2]);
            f3 e2 = mk(te2[3 * i], te2[3 * i + 1], te2[3 * i + 2]);
            f3 c21 = cross(e2, e1);
            f3 q = L - p;
            f3 ceq = cross(e2, q), cqe = cross(q, e1);
            float nc = sqrtf(dot(c21, c21)), nu = sqrtf(dot(ceq, ceq)), nv = sqrtf(dot(cqe, cqe));
            float ne = sqrtf(dot(e1, e1)) + sqrtf(dot(e2, e2));
            float slack = FILTER_KAPPA * ((nc + nu + nv) + ne * (sqrtf(dot(L, L)) + abs_bound));
            if (!(slack < 3.0e38f)) { cu = zero; cv = zero; }
            else if (nc > 0.f) {
                float k = 1.0f / slack;
                cu = ceq * k; cv = cqe * k; cw = ((c21 - ceq) - cqe) * k;
            }
        }
        set_cam_half(pc, h, cu, cv, cw);
    }
    pair_lamp[j] = pc;
}

// ---------------------------------------------------------------------------------------
// Chunk culling.  Triangles are scanned in chunks of CHUNK_TRIS consecutive indices; a chunk
// whose (inflated) bounding box the half-line of a ray misses cannot contain a hit of that
// ray, so a wave skips the chunk when that holds for all its rays.  Chunks are still visited
// in index order, which keeps the lowest-index tie rule and the first-occluder order.
//
// The inflation makes the skip exact with respect to the EXACT tests, which accept hits
// slightly outside a triangle: the barycentrics carry an absolute error of at most
// 8*2^-24*|d||e||q| / |det| and the tests require |det| > 1e-4, so an accepted hit lies within
// 9.6e-3*|d|*|e1||e2|*|q| of its triangle.  A = 2e-2 * max |e1||e2| over the chunk (2x margin),
// |q| <= |o - centre|_1 + |half extent|_1, and e_abs covers the rounding of the slab test.
// ---------------------------------------------------------------------------------------
constexpr int CHUNK_TRIS = 2 * LDS_CHUNK;

__global__ void k_setup_chunks(uint32_t n_tri, uint32_t n_chunks, const float *tp, const float *te1, const float *te2,
                               float4 *chunk_box)
{
    uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chunks) return;
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    float amax = 0.f; bool finite = true, any = false;
    for (uint32_t i = c * CHUNK_TRIS; i < (c + 1) * CHUNK_TRIS && i < n_tri; i++) {
        any = true;
        float e1l = 0.f, e2l = 0.f;
        for (int k = 0; k < 3; k++) {
            float p = tp[3 * i + k], a = te1[3 * i + k], b = te2[3 * i + k];
            float v1 = p + a, v2 = p + b;
            lo[k] = fminf(lo[k], fminf(p, fminf(v1, v2))); hi[k] = fmaxf(hi[k], fmaxf(p, fmaxf(v1, v2)));
            e1l += a * a; e2l += b * b;
            finite = finite && (fabsf(p) < 3.0e38f) && (fabsf(v1) < 3.0e38f) && (fabsf(v2) < 3.0e38f);
        }
        amax = fmaxf(amax, sqrtf(e1l) * sqrtf(e2l));
    }
    float4 b0, b1, b2;
    if (!any) {                       // padding only: nothing to hit
        b0 = make_float4(0.f, 0.f, 0.f, 0.f); b1 = make_float4(-1.f, -1.f, -1.f, 0.f); b2 = make_float4(0.f, 0.f, 0.f, 0.f);
    } else if (!finite || !(amax < 1.0e30f)) {   // never skip
        b0 = make_float4(0.f, 0.f, 0.f, 0.f); b1 = make_float4(3.0e37f, 3.0e37f, 3.0e37f, 0.f); b2 = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
        float cx = 0.5f * (lo[0] + hi[0]), cy = 0.5f * (lo[1] + hi[1]), cz = 0.5f * (lo[2] + hi[2]);
        float hx = 0.5f * (hi[0] - lo[0]), hy = 0.5f * (hi[1] - lo[1]), hz = 0.5f * (hi[2] - lo[2]);
        float mag = fabsf(cx) + fabsf(cy) + fabsf(cz) + hx + hy + hz;
        float A = 2.0e-2f * amax;
        float eabs = 1.0e-5f * mag + 1.0e-30f;
        b0 = make_float4(cx, cy, cz, A);
        b1 = make_float4(hx * 1.00001f, hy * 1.00001f, hz * 1.00001f, eabs);
        b2 = make_float4(A * (hx + hy + hz) * 1.00001f, 0.f, 0.f, 0.f);
    }
    chunk_box[3 * c] = b0; chunk_box[3 * c + 1] = b1; chunk_box[3 * c + 2] = b2;
}

// per-ray constants of the slab test: origin, clamped reciprocal direction, |d|
struct RayBox { f3 o, inv, ainv; float dlen; };
DEV float safe_rcp(float x) {
    float r = 1.0f / x;
    if (!(fabsf(r) < 1.0e28f)) r = (__float_as_uint(x) >> 31) ? -1.0e28f : 1.0e28f;   // |x| tiny, zero or NaN
    return r;
}
DEV RayBox raybox_make(f3 o, f3 d) {
    RayBox r;
    r.o = o;
    r.inv = mk(safe_rcp(d.x), safe_rcp(d.y), safe_rcp(d.z));
    r.ainv = mk(fabsf(r.inv.x), fabsf(r.inv.y), fabsf(r.inv.z));
    r.dlen = sqrtf(dot(d, d));
    if (!(r.dlen < 3.0e38f)) r.dlen = 3.0e38f;
    return r;
}
// true unless the half-line {o + t d, t >= 0} certainly misses the inflated box of the chunk
DEV bool chunk_may_hit(const RayBox &r, float4 b0, float4 b1, float4 b2) {
    float mx = b0.x - r.o.x, my = b0.y - r.o.y, mz = b0.z - r.o.z;
    float m1 = (fabsf(mx) + fabsf(my)) + fabsf(mz);
    float infl = fmaf(b0.w * r.dlen, m1, fmaf(b2.x, r.dlen, b1.w));
    float hx = b1.x + infl, hy = b1.y + infl, hz = b1.z + infl;
    float tcx = mx * r.inv.x, tcy = my * r.inv.y, tcz = mz * r.inv.z;
    float thx = hx * r.ainv.x, thy = hy * r.ainv.y, thz = hz * r.ainv.z;
    float tmin = fmaxf(fmaxf(tcx - thx, tcy - thy), tcz - thz);
    float tmax = fminf(fminf(tcx + thx, tcy + thy), tcz + thz);
    return !(tmax < fmaxf(tmin, 0.0f)) || !(infl < 3.0e37f);
}

// ---------------------------------------------------------------------------------------
// FILTER tests (packed FP32: one instruction evaluates both triangles of a pair)
// ---------------------------------------------------------------------------------------
DEV v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
DEV v2f splat(float s) { v2f r; r.x = s; r.y = s; return r; }
DEV float min3f(float a, float b, float c) { return fminf(fminf(a, b), c); }
DEV float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
#define RT_HALF(v, h) ((h) ? (v).y : (v).x)

// the three edge functions of both triangles of a pair for one direction (as splat pairs)
DEV void edges_cam_pair(const PairCam &P, v2f Dx, v2f Dy, v2f Dz, v2f &su, v2f &sv, v2f &sw) {
    su = pk_fma(P.uz, Dz, pk_fma(P.uy, Dy, P.ux * Dx));
    sv = pk_fma(P.vz, Dz, pk_fma(P.vy, Dy, P.vx * Dx));
    sw = pk_fma(P.wz, Dz, pk_fma(P.wy, Dy, P.wx * Dx));
}
DEV void edges_cam_one(const PairCam &P, int h, f3 d, float &su, float &sv, float &sw) {
    su = fmaf(RT_HALF(P.uz, h), d.z, fmaf(RT_HALF(P.uy, h), d.y, RT_HALF(P.ux, h) * d.x));
    sv = fmaf(RT_HALF(P.vz, h), d.z, fmaf(RT_HALF(P.vy, h), d.y, RT_HALF(P.vx, h) * d.x));
    sw = fmaf(RT_HALF(P.wz, h), d.z, fmaf(RT_HALF(P.wy, h), d.y, RT_HALF(P.wx, h) * d.x));
}
// camera rays (one-sided): acc = max(acc, min(u,v,w)) per triangle; may hit iff acc >= -theta
DEV float filter_cam_pair(const PairCam &P, v2f Dx, v2f Dy, v2f Dz, float acc) {
    v2f su, sv, sw;
    edges_cam_pair(P, Dx, Dy, Dz, su, sv, sw);
    return max3f(acc, min3f(su.x, sv.x, sw.x), min3f(su.y, sv.y, sw.y));
}
DEV float filter_cam_one(const PairCam &P, int h, f3 d) {
    float su, sv, sw;
    edges_cam_one(P, h, d, su, sv, sw);
    return min3f(su, sv, sw);
}
// lines through a lamp (two-sided): acc = max(acc, max(min(u,v,w), -max(u,v,w)))
DEV float filter_line_pair(const PairCam &P, v2f Dx, v2f Dy, v2f Dz, float acc) {
    v2f su, sv, sw;
    edges_cam_pair(P, Dx, Dy, Dz, su, sv, sw);
    acc = max3f(acc, min3f(su.x, sv.x, sw.x), -max3f(su.x, sv.x, sw.x));
    return max3f(acc, min3f(su.y, sv.y, sw.y), -max3f(su.y, sv.y, sw.y));
}
DEV float filter_line_one(const PairCam &P, int h, f3 d) {
    float su, sv, sw;
    edges_cam_one(P, h, d, su, sv, sw);
    return fmaxf(min3f(su, sv, sw), -max3f(su, sv, sw));
}

// per-ray constants of the general FILTER: unit direction, moment about `center`, threshold
struct RayF { v2f D[3], M[3]; f3 d, mo; float thr; };
DEV RayF rayf_make(const DScene &S, f3 o, f3 d) {
    RayF r;
    f3 op = mk(o.x - S.center[0], o.y - S.center[1], o.z - S.center[2]);
    float inv = 1.0f / sqrtf(dot(d, d));
    r.d = d * inv;
    r.mo = cross(op, r.d);
    float far = sqrtf(dot(op, op)) / S.rmax;
    r.thr = -(S.theta * fmaxf(1.0f, far));
    if (!(r.thr > -3.0e38f) || !(inv < 3.0e38f) || !(inv > 0.f)) {   // degenerate ray: exact test decides
        r.thr = -3.0e38f; r.d = mk(0.f, 0.f, 0.f); r.mo = mk(0.f, 0.f, 0.f);
    }
    r.D[0] = splat(r.d.x); r.D[1] = splat(r.d.y); r.D[2] = splat(r.d.z);
    r.M[0] = splat(r.mo.x); r.M[1] = splat(r.mo.y); r.M[2] = splat(r.mo.z);
    return r;
}
// acc = max(acc, max(min(u,v,w), -max(u,v,w))) per triangle of the pair; may hit iff acc >= thr
DEV float filter_gen_pair(const PairGen &P, const RayF &r, float acc) {
    v2f su = pk_fma(P.um[2], r.M[2], pk_fma(P.um[1], r.M[1], pk_fma(P.um[0], r.M[0],
             pk_fma(P.ud[2], r.D[2], pk_fma(P.ud[1], r.D[1], P.ud[0] * r.D[0])))));
    v2f sv = pk_fma(P.vm[2], r.M[2], pk_fma(P.vm[1], r.M[1], pk_fma(P.vm[0], r.M[0],
             pk_fma(P.vd[2], r.D[2], pk_fma(P.vd[1], r.D[1], P.vd[0] * r.D[0])))));
    v2f sw = pk_fma(P.wm[2], r.M[2], pk_fma(P.wm[1], r.M[1], pk_fma(P.wm[0], r.M[0],
             pk_fma(P.wd[2], r.D[2], pk_fma(P.wd[1], r.D[1], P.wd[0] * r.D[0])))));
    acc = max3f(acc, min3f(su.x, sv.x, sw.x), -max3f(su.x, sv.x, sw.x));
    return max3f(acc, min3f(su.y, sv.y, sw.y), -max3f(su.y, sv.y, sw.y));
}
DEV float filter_gen_one(const PairGen &P, int h, const RayF &r) {
    float su = fmaf(RT_HALF(P.um[2], h), r.mo.z, fmaf(RT_HALF(P.um[1], h), r.mo.y, fmaf(RT_HALF(P.um[0], h), r.mo.x,
               fmaf(RT_HALF(P.ud[2], h), r.d.z, fmaf(RT_HALF(P.ud[1], h), r.d.y, RT_HALF(P.ud[0], h) * r.d.x)))));
    float sv = fmaf(RT_HALF(P.vm[2], h), r.mo.z, fmaf(RT_HALF(P.vm[1], h), r.mo.y, fmaf(RT_HALF(P.vm[0], h), r.mo.x,
               fmaf(RT_HALF(P.vd[2], h), r.d.z, fmaf(RT_HALF(P.vd[1], h), r.d.y, RT_HALF(P.vd[0], h) * r.d.x)))));
    float sw = fmaf(RT_HALF(P.wm[2], h), r.mo.z, fmaf(RT_HALF(P.wm[1], h), r.mo.y, fmaf(RT_HALF(P.wm[0], h), r.mo.x,
               fmaf(RT_HALF(P.wd[2], h), r.d.z, fmaf(RT_HALF(P.wd[1], h), r.d.y, RT_HALF(P.wd[0], h) * r.d.x)))));
    return fmaxf(min3f(su, sv, sw), -max3f(su, sv, sw));
}

// ---------------------------------------------------------------------------------------
// k_primary
// ---------------------------------------------------------------------------------------
DEV bool slot_to_pixel(const Frame &F, uint32_t slot, int &x, int &y, int &lrow) {
    uint32_t blk = slot >> 6, lane = slot & 63;
    int k = (int)(blk / (uint32_t)F.nbx), bx = (int)(blk % (uint32_t)F.nbx);
    x = (F.x_blk0 + bx) * 8 + (int)(lane & 7);
    int ly = (int)(lane >> 3);
    lrow = (F.out_first + k * F.out_step) * HIPEXEC_TILE_ROWS + ly;
    y = (F.tile_first + k * F.tile_step) * HIPEXEC_TILE_ROWS + ly;
    return slot < F.n_slots && x >= F.cx0 && x < F.cx1 && y >= F.cy0 && y < F.cy1;
}

// Sample index of sub-sample r of pixel slot `slot` (slots: 64 per 8x8-pixel tile, row-major inside the tile).
// With 4 sub-samples, 64 consecutive indices are the 4 x 4 sub-samples of one 4x4-pixel quadrant: the rays of a
// wave then differ by at most 4 pixels instead of 8, which is what the packet walk of the camera rays wants
// (a third fewer steps per wave), and a block of 256 is one tile with all its sub-samples.
template <int SPP>
DEV uint32_t sample_index(const Frame &F, uint32_t slot, int r) {
    if (SPP == 1) return slot;
    const uint32_t blk = slot >> 6, lane = slot & 63u, lx = lane & 7u, ly = lane >> 3;
    const uint32_t quad = (ly >> 2) * 2u + (lx >> 2), pix = (ly & 3u) * 4u + (lx & 3u);
    return ((blk * 4u + quad) * 4u + (uint32_t)r) * 16u + pix;
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

// direction of sub-sample r of pixel (x, y) alone: the expressions of camera_rays for d[r]
template <int MODE>
DEV f3 camera_ray(const DScene &S, const Frame &F, int x, int y, int r) {
    float camHeight = 0.5f;
    float camWidth = (float)F.W * (camHeight / (float)F.H);
    float camDist = 1.0f;
    float dh = camHeight / (float)F.H;
    float dw = camWidth / (float)F.W;
    float rayX = -(camWidth / 2) + (float)x * dw;
    float rayY = -(camHeight / 2) + (float)y * dh;
    f3 world = vecmat(mk(rayX, rayY, -camDist), S.cam_mat);
    if (MODE != HIPEXEC_MODE_CL) {
        f3 wdx = vecmat(mk(dw / 2.0f, 0.0f, 0.0f), S.cam_mat);
        f3 wdy = vecmat(mk(0.0f, dh / 2.0f, 0.0f), S.cam_mat);
        if (r == 1) world = world + wdx;
        else if (r == 2) world = world + wdy;
        else if (r == 3) world = (world + wdx) + wdy;
    }
    return normalize(world);
}

template <int MODE>
__global__ __launch_bounds__(256) void k_primary(DScene S, Frame F, float4 *__restrict__ rays, float4 *__restrict__ hits,
                                                 unsigned long long *__restrict__ counters)
{
    uint32_t n_groups = 0;
    constexpr int SPP = (MODE == HIPEXEC_MODE_CL) ? 1 : 4;
    constexpr int UP = 2;                                   // triangle pairs per loop iteration
    uint32_t slot = blockIdx.x * 256u + threadIdx.x;
    int x, y, lrow;
    bool active = slot_to_pixel(F, slot, x, y, lrow);
    if (!active) { x = min(x, F.W - 1); y = min(y, F.H - 1); }     // keep idle lanes on finite data
    f3 d[SPP];
    camera_rays<MODE>(S, F, x, y, d);
    v2f Dx[SPP], Dy[SPP], Dz[SPP];
    Best tb[SPP];
#pragma unroll
    for (int r = 0; r < SPP; r++) { tb[r] = best_none(); Dx[r] = splat(d[r].x); Dy[r] = splat(d[r].y); Dz[r] = splat(d[r].z); }
    const float thr = -S.theta;
    const PairCam *__restrict__ PC = S.pair_cam;
    const float4 *__restrict__ PX = S.tri_primx;
    const float4 *__restrict__ CB = S.chunk_box;
    const uint32_t n_chunks = (S.n_tri_pad / 2) / LDS_CHUNK;
    RayBox rb[SPP];
    uint32_t n_scanned = 0;
#pragma unroll
    for (int r = 0; r < SPP; r++) rb[r] = raybox_make(mk(S.cam_pos[0], S.cam_pos[1], S.cam_pos[2]), d[r]);
    for (uint32_t c = 0; c < n_chunks; c++) {
      if (S.cull) {
        float4 b0 = CB[3 * c], b1 = CB[3 * c + 1], b2 = CB[3 * c + 2];
        bool may = false;
#pragma unroll
        for (int r = 0; r < SPP; r++) may |= chunk_may_hit(rb[r], b0, b1, b2);
        if (__ballot(may) == 0) continue;
      }
      n_scanned++;
      for (uint32_t jp = c * LDS_CHUNK; jp < (c + 1) * LDS_CHUNK; jp += UP) {
        float acc[SPP];
#pragma unroll
        for (int r = 0; r < SPP; r++) acc[r] = -3.0e38f;
#pragma unroll
        for (int k = 0; k < UP; k++) {
            PairCam P = PC[jp + k];
#pragma unroll
            for (int r = 0; r < SPP; r++) acc[r] = filter_cam_pair(P, Dx[r], Dy[r], Dz[r], acc[r]);
        }
        bool any = false;
#pragma unroll
        for (int r = 0; r < SPP; r++) any |= !(acc[r] < thr);
        if (__ballot(any)) {
            n_groups++;
#pragma unroll 1
            for (uint32_t ti = 2 * jp; ti < 2 * (jp + UP); ti++) {
                PairCam P = PC[ti >> 1];
                int h = (int)(ti & 1u);
                float4 e0 = PX[3 * ti], e1 = PX[3 * ti + 1], e2 = PX[3 * ti + 2];
#pragma unroll
                for (int r = 0; r < SPP; r++) {
                    if (!(filter_cam_one(P, h, d[r]) < thr)) {
                        float t, u, v;
                        if (ti < S.n_tri && tri_exact_cam<MODE>(d[r], xyz(e0), e0.w, xyz(e1), xyz(e2), t, u, v))
                            best_update<MODE>(tb[r], t, u, v, (int)ti);
                    }
                }
            }
        }
      }
    }
    if ((threadIdx.x & 63u) == 0) {
        if (n_groups) atomicAdd(&counters[2], (unsigned long long)n_groups);
        if (n_scanned) atomicAdd(&counters[5], (unsigned long long)n_scanned);
    }
    f3 o = mk(S.cam_pos[0], S.cam_pos[1], S.cam_pos[2]);
#pragma unroll
    for (int r = 0; r < SPP; r++) {
        Best sb = spheres_closest<MODE>(S, o, d[r]);
        float4 h = merge_hits<MODE>(S, tb[r], sb);
        if (!active) h.w = __int_as_float(-2);            // pixel slot outside the frame
        uint32_t s = sample_index<SPP>(F, slot, r);
        if (slot < F.n_slots) {
            hits[s] = h;
            store_ray(rays, s, o, d[r], s, 0u);
        }
    }
}

// ---------------------------------------------------------------------------------------
// k_closest: closest hit of n queued rays, R rays per lane
// ---------------------------------------------------------------------------------------
template <int MODE, int R, bool LDS>
__global__ __launch_bounds__(256) void k_closest(DScene S, const float4 *__restrict__ rays, uint32_t n,
                                                 float4 *__restrict__ hits, unsigned long long *__restrict__ counters)
{
    constexpr int UP = 2;
    constexpr int F4 = LDS_CHUNK * (int)(sizeof(PairGen) / 16);      // float4 per chunk (576)
    __shared__ float4 lds[LDS ? 2 : 1][LDS ? F4 : 1];
    uint32_t n_groups = 0;
    const uint32_t tid = threadIdx.x;
    uint32_t wave = (blockIdx.x * 256u + tid) >> 6;
    uint32_t lane = tid & 63u;
    uint32_t base = wave * 64u * R + lane;
    f3 o[R], d[R]; RayF rf[R]; RayBox rb[R]; Best tb[R]; bool valid[R];
    uint32_t n_scanned = 0;
#pragma unroll
    for (int r = 0; r < R; r++) {
        uint32_t j = base + 64u * r;
        valid[r] = j < n;
        uint32_t s, aux;
        load_ray(rays, valid[r] ? j : 0u, o[r], d[r], s, aux);
        rf[r] = rayf_make(S, o[r], d[r]);
        rb[r] = raybox_make(o[r], d[r]);
        tb[r] = best_none();
    }
    const PairGen *__restrict__ PG = S.pair_gen;
    const float4 *__restrict__ EX = S.tri_exact;
    const float4 *__restrict__ CB = S.chunk_box;
    const float4 *__restrict__ G = reinterpret_cast<const float4 *>(PG);
    const uint32_t n_chunks = (S.n_tri_pad / 2) / LDS_CHUNK;
    if (LDS) {
        if (n_chunks) for (int k = 0; k < (F4 + 255) / 256; k++) if (tid + 256 * k < F4) lds[0][tid + 256 * k] = G[tid + 256 * k];
        __syncthreads();
    }
    for (uint32_t c = 0; c < n_chunks; c++) {
        const int buf = LDS ? (int)(c & 1u) : 0;
        float4 pre[(F4 + 255) / 256];
        const bool more = c + 1 < n_chunks;
        if (LDS && more) {
            const float4 *g = G + (size_t)(c + 1) * F4;
#pragma unroll
            for (int k = 0; k < (F4 + 255) / 256; k++) if (tid + 256 * k < F4) pre[k] = g[tid + 256 * k];
        }
        const PairGen *L = LDS ? reinterpret_cast<const PairGen *>(&lds[buf][0]) : (PG + (size_t)c * LDS_CHUNK);
        bool scan = true;
        if (S.cull) {
            float4 b0 = CB[3 * c], b1 = CB[3 * c + 1], b2 = CB[3 * c + 2];
            bool may = false;
#pragma unroll
            for (int r = 0; r < R; r++) may |= valid[r] && chunk_may_hit(rb[r], b0, b1, b2);
            scan = __ballot(may) != 0;
        }
        if (scan) n_scanned++;
        for (uint32_t jl = 0; scan && jl < LDS_CHUNK; jl += UP) {
            float acc[R];
#pragma unroll
            for (int r = 0; r < R; r++) acc[r] = -3.0e38f;
#pragma unroll
            for (int k = 0; k < UP; k++) {
                PairGen P = L[jl + k];
#pragma unroll
                for (int r = 0; r < R; r++) acc[r] = filter_gen_pair(P, rf[r], acc[r]);
            }
            bool any = false;
#pragma unroll
            for (int r = 0; r < R; r++) any |= !(acc[r] < rf[r].thr);
            if (__ballot(any)) {
                n_groups++;
#pragma unroll 1
                for (uint32_t tl = 2 * jl; tl < 2 * (jl + UP); tl++) {
                    uint32_t ti = 2 * c * LDS_CHUNK + tl;
                    PairGen P = L[tl >> 1];
                    int h = (int)(tl & 1u);
                    float4 e0 = EX[3 * ti], e1 = EX[3 * ti + 1], e2 = EX[3 * ti + 2];
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        if (!(filter_gen_one(P, h, rf[r]) < rf[r].thr)) {
                            float t, u, v;
                            if (ti < S.n_tri && tri_exact<MODE>(o[r], d[r], xyz(e0), xyz(e1), xyz(e2), t, u, v))
                                best_update<MODE>(tb[r], t, u, v, (int)ti);
                        }
                    }
                }
            }
        }
        if (LDS) {
            if (more) {
#pragma unroll
                for (int k = 0; k < (F4 + 255) / 256; k++) if (tid + 256 * k < F4) lds[buf ^ 1][tid + 256 * k] = pre[k];
            }
            __syncthreads();
        }
    }
    if (lane == 0) {
        if (n_groups) atomicAdd(&counters[3], (unsigned long long)n_groups);
        if (n_scanned) atomicAdd(&counters[6], (unsigned long long)n_scanned);
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
template <int MODE, int R>
__global__ __launch_bounds__(256) void k_anyhit(DScene S, const float4 *__restrict__ rays, uint32_t n,
                                                uint32_t *__restrict__ occl, uint32_t lw,
                                                unsigned long long *__restrict__ counters)
{
    constexpr int UP = 2;
    uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6;
    uint32_t lane = threadIdx.x & 63u;
    uint32_t base = wave * 64u * R + lane;
    uint32_t n_groups = 0;
    f3 o[R], d[R]; RayF rf[R]; bool done[R], occluded[R]; uint32_t s[R], lamp[R], tri_tests[R], sph_tests[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        uint32_t j = base + 64u * r;
        bool valid = j < n;
        load_ray(rays, valid ? j : 0u, o[r], d[r], s[r], lamp[r]);
        rf[r] = rayf_make(S, o[r], d[r]);
        done[r] = !valid; occluded[r] = false;
        tri_tests[r] = valid ? S.n_tri : 0u; sph_tests[r] = valid ? S.n_sph : 0u;
    }
    const PairGen *__restrict__ PG = S.pair_gen;
    const float4 *__restrict__ EX = S.tri_exact;
    const float4 *__restrict__ CB = S.chunk_box;
    const uint32_t n_chunks = (S.n_tri_pad / 2) / LDS_CHUNK;
    RayBox rb[R];
    uint32_t n_scanned = 0;
#pragma unroll
    for (int r = 0; r < R; r++) rb[r] = raybox_make(o[r], d[r]);
    for (uint32_t c = 0; c < n_chunks; c++) {
      {
        bool alive = false;
#pragma unroll
        for (int r = 0; r < R; r++) alive |= !done[r];
        if (__ballot(alive) == 0) break;
      }
      if (S.cull) {
        float4 b0 = CB[3 * c], b1 = CB[3 * c + 1], b2 = CB[3 * c + 2];
        bool may = false;
#pragma unroll
        for (int r = 0; r < R; r++) may |= !done[r] && chunk_may_hit(rb[r], b0, b1, b2);
        if (__ballot(may) == 0) continue;
      }
      n_scanned++;
      for (uint32_t jp = c * LDS_CHUNK; jp < (c + 1) * LDS_CHUNK; jp += UP) {
        float acc[R];
#pragma unroll
        for (int r = 0; r < R; r++) acc[r] = -3.0e38f;
#pragma unroll
        for (int k = 0; k < UP; k++) {
            PairGen P = PG[jp + k];
#pragma unroll
            for (int r = 0; r < R; r++) acc[r] = filter_gen_pair(P, rf[r], acc[r]);
        }
        bool any = false;
#pragma unroll
        for (int r = 0; r < R; r++) any |= !(acc[r] < rf[r].thr) && !done[r];
        if (__ballot(any)) {
            n_groups++;
#pragma unroll 1
            for (uint32_t ti = 2 * jp; ti < 2 * (jp + UP); ti++) {
                PairGen P = PG[ti >> 1];
                int h = (int)(ti & 1u);
                float4 e0 = EX[3 * ti], e1 = EX[3 * ti + 1], e2 = EX[3 * ti + 2];
#pragma unroll
                for (int r = 0; r < R; r++) {
                    if (!done[r] && ti < S.n_tri && !(filter_gen_one(P, h, rf[r]) < rf[r].thr)) {
                        float t, u, v;
                        if (tri_exact<MODE>(o[r], d[r], xyz(e0), xyz(e1), xyz(e2), t, u, v)) {
                            done[r] = true; occluded[r] = true; tri_tests[r] = ti + 1; sph_tests[r] = 0;
                        }
                    }
                }
            }
        }
      }
    }
    unsigned long long a = 0, b = 0;
#pragma unroll
    for (int r = 0; r < R; r++) {
        if (!done[r]) {
            for (uint32_t k = 0; k < S.n_sph; k++) {
                float4 sp = S.sph[k];
                float t;
                if (!done[r] && sph_exact<MODE>(o[r], d[r], xyz(sp), sp.w, t)) { done[r] = true; occluded[r] = true; sph_tests[r] = k + 1; }
            }
        }
        if (occluded[r]) atomicOr(&occl[(size_t)s[r] * lw + (lamp[r] >> 5)], 1u << (lamp[r] & 31u));
        a += tri_tests[r]; b += sph_tests[r];
    }
    a = wave_sum_u64(a); b = wave_sum_u64(b);
    if (lane == 0) {
        if (a) atomicAdd(&counters[0], a);
        if (b) atomicAdd(&counters[1], b);
        if (n_groups) atomicAdd(&counters[4], (unsigned long long)n_groups);
        if (n_scanned) atomicAdd(&counters[7], (unsigned long long)n_scanned);
    }
}

// ---------------------------------------------------------------------------------------
// k_anyhit_lamp: the same occlusion query for shadow rays that all point at one lamp, with the
// cheaper lamp-line FILTER (pair_lamp records of that lamp).  Exact tests and results are
// those of k_anyhit.
// ---------------------------------------------------------------------------------------
template <int MODE, int R, bool LDS>
__global__ __launch_bounds__(256) void k_anyhit_lamp(DScene S, const PairCam *__restrict__ PL, const float4 *__restrict__ rays,
                                                     uint32_t n, uint32_t *__restrict__ occl, uint32_t lw,
                                                     unsigned long long *__restrict__ counters)
{
    constexpr int UP = 2;
    constexpr int F4 = LDS_CHUNK * (int)(sizeof(PairCam) / 16);      // float4 per chunk (320)
    __shared__ float4 lds[LDS ? 2 : 1][LDS ? F4 : 1];
    const uint32_t tid = threadIdx.x;
    uint32_t wave = (blockIdx.x * 256u + tid) >> 6;
    uint32_t lane = tid & 63u;
    uint32_t base = wave * 64u * R + lane;
    uint32_t n_groups = 0;
    f3 o[R], d[R], dn[R]; v2f Dx[R], Dy[R], Dz[R]; RayBox rb[R];
    bool done[R], occluded[R]; uint32_t s[R], lamp[R], tri_tests[R], sph_tests[R];
    float thr[R];
    uint32_t n_scanned = 0;
    const float4 *__restrict__ CB = S.chunk_box;
#pragma unroll
    for (int r = 0; r < R; r++) {
        uint32_t j = base + 64u * r;
        bool valid = j < n;
        load_ray(rays, valid ? j : 0u, o[r], d[r], s[r], lamp[r]);
        float inv = 1.0f / sqrtf(dot(d[r], d[r]));
        dn[r] = d[r] * inv;
        thr[r] = -S.theta;
        if (!(inv < 3.0e38f) || !(inv > 0.f) || !(thr[r] > -3.0e38f)) { dn[r] = mk(0.f, 0.f, 0.f); thr[r] = -3.0e38f; }
        Dx[r] = splat(dn[r].x); Dy[r] = splat(dn[r].y); Dz[r] = splat(dn[r].z);
        rb[r] = raybox_make(o[r], d[r]);
        done[r] = !valid; occluded[r] = false;
        tri_tests[r] = valid ? S.n_tri : 0u; sph_tests[r] = valid ? S.n_sph : 0u;
    }
    const float4 *__restrict__ EX = S.tri_exact;
    const float4 *__restrict__ G = reinterpret_cast<const float4 *>(PL);
    const uint32_t n_chunks = (S.n_tri_pad / 2) / LDS_CHUNK;
    if (LDS) {
        if (n_chunks) {
            lds[0][tid] = G[tid];
            if (tid < F4 - 256) lds[0][256 + tid] = G[256 + tid];
        }
        __syncthreads();
    }
    for (uint32_t c = 0; c < n_chunks; c++) {
        const int buf = LDS ? (int)(c & 1u) : 0;
        float4 pre0 = make_float4(0.f, 0.f, 0.f, 0.f), pre1 = pre0;
        const bool more = c + 1 < n_chunks;
        if (LDS && more) {
            const float4 *g = G + (size_t)(c + 1) * F4;
            pre0 = g[tid];
            if (tid < F4 - 256) pre1 = g[256 + tid];
        }
        bool alive = false;
#pragma unroll
        for (int r = 0; r < R; r++) alive |= !done[r];
        bool scan = __ballot(alive) != 0;
        if (scan && S.cull) {
            float4 b0 = CB[3 * c], b1 = CB[3 * c + 1], b2 = CB[3 * c + 2];
            bool may = false;
#pragma unroll
            for (int r = 0; r < R; r++) may |= !done[r] && chunk_may_hit(rb[r], b0, b1, b2);
            scan = __ballot(may) != 0;
        }
        if (scan) {
            n_scanned++;
            const PairCam *L = LDS ? reinterpret_cast<const PairCam *>(&lds[buf][0]) : (PL + (size_t)c * LDS_CHUNK);
            for (uint32_t jl = 0; jl < LDS_CHUNK; jl += UP) {
                float acc[R];
#pragma unroll
                for (int r = 0; r < R; r++) acc[r] = -3.0e38f;
#pragma unroll
                for (int k = 0; k < UP; k++) {
                    PairCam P = L[jl + k];
#pragma unroll
                    for (int r = 0; r < R; r++) acc[r] = filter_line_pair(P, Dx[r], Dy[r], Dz[r], acc[r]);
                }
                bool any = false;
#pragma unroll
                for (int r = 0; r < R; r++) any |= !(acc[r] < thr[r]) && !done[r];
                if (__ballot(any)) {
                    n_groups++;
#pragma unroll 1
                    for (uint32_t tl = 2 * jl; tl < 2 * (jl + UP); tl++) {
                        uint32_t ti = 2 * c * LDS_CHUNK + tl;
                        PairCam P = L[tl >> 1];
                        int h = (int)(tl & 1u);
                        float4 e0 = EX[3 * ti], e1 = EX[3 * ti + 1], e2 = EX[3 * ti + 2];
#pragma unroll
                        for (int r = 0; r < R; r++) {
                            if (!done[r] && ti < S.n_tri && !(filter_line_one(P, h, dn[r]) < thr[r])) {
                                float t, u, v;
                                if (tri_exact<MODE>(o[r], d[r], xyz(e0), xyz(e1), xyz(e2), t, u, v)) {
                                    done[r] = true; occluded[r] = true; tri_tests[r] = ti + 1; sph_tests[r] = 0;
                                }
                            }
                        }
                    }
                }
            }
        }
        alive = false;
#pragma unroll
        for (int r = 0; r < R; r++) alive |= !done[r];
        if (LDS) {
            if (more) {
                lds[buf ^ 1][tid] = pre0;
                if (tid < F4 - 256) lds[buf ^ 1][256 + tid] = pre1;
            }
            if (!__syncthreads_or(alive ? 1 : 0)) break;
        } else if (__ballot(alive) == 0) break;
    }
    unsigned long long a = 0, b = 0;
#pragma unroll
    for (int r = 0; r < R; r++) {
        if (!done[r]) {
            for (uint32_t k = 0; k < S.n_sph; k++) {
                float4 sp = S.sph[k];
                float t;
                if (!done[r] && sph_exact<MODE>(o[r], d[r], xyz(sp), sp.w, t)) { done[r] = true; occluded[r] = true; sph_tests[r] = k + 1; }
            }
        }
        if (occluded[r]) atomicOr(&occl[(size_t)s[r] * lw + (lamp[r] >> 5)], 1u << (lamp[r] & 31u));
        a += tri_tests[r]; b += sph_tests[r];
    }
    a = wave_sum_u64(a); b = wave_sum_u64(b);
    if (lane == 0) {
        if (a) atomicAdd(&counters[0], a);
        if (b) atomicAdd(&counters[1], b);
        if (n_groups) atomicAdd(&counters[4], (unsigned long long)n_groups);
        if (n_scanned) atomicAdd(&counters[7], (unsigned long long)n_scanned);
    }
}

// ---------------------------------------------------------------------------------------
// k_shadowgen: one thread per ray of the current level
// ---------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void k_shadowgen(DScene S, const float4 *__restrict__ rays, const float4 *__restrict__ hits,
                                                   uint32_t n, float4 *__restrict__ shadow_q, uint32_t *__restrict__ q_count)
{
    // queue of lamp l: records [l*n, l*n + q_count[l])
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
        uint32_t pos = wave_append(cast, q_count + l);
        if (cast) store_ray(shadow_q + 2 * (size_t)l * n, pos, sf.P, toLamp, s, l);
    }
}

// ---------------------------------------------------------------------------------------
// Colour of one sample of a level: {colour, w} with w = reflectionFactor if the sample continues
// with a bounce ray (returned in bo, bd) and -1 otherwise (ray_tracer.cpp:165-199 /
// ray_tracer_cl.cpp:119-174).  `occ` points at the sample's occlusion words (bit l: lamp l is hidden).
// ---------------------------------------------------------------------------------------
template <int MODE>
DEV float4 shade_sample(const DScene &S, f3 o, f3 d, float4 h, const uint32_t *occ, bool spawn, bool &bounce, f3 &bo, f3 &bd,
                        float ao = 0.0f)
{
    f3 ret; float w = -1.0f;
    if (__float_as_int(h.w) < 0) {
        // miss: ray_tracer.cpp:199 returns the horizon; ray_tracer_cl.cpp:119-123 adds it to 0
        f3 hz = mk(S.horizon[0], S.horizon[1], S.horizon[2]);
        ret = (MODE == HIPEXEC_MODE_CL) ? (mk(0.f, 0.f, 0.f) + hz) : hz;
    } else {
        Surf sf = surface(S, o, d, h, true);
        ret = (MODE == HIPEXEC_MODE_CL) ? mk(0.f, 0.f, 0.f) : mk(S.ambient[0], S.ambient[1], S.ambient[2]);
        if (MODE == HIPEXEC_MODE_CPU && S.ao) ret = ret + sf.color * ao;           // ray_tracer.cpp:167-169
        for (uint32_t l = 0; l < S.n_lamp; l++) {
            float4 lp = S.lamp[l];
            f3 toLamp = xyz(lp) - sf.P;
            float dotWithLamp = dot(toLamp, sf.N);
            float dotWithDir = dot(d, sf.N);
            if (!casts_shadow_ray<MODE>(dotWithDir, dotWithLamp)) continue;
            if (occ[l >> 5] & (1u << (l & 31u))) continue;
            ret = add_lamp<MODE>(ret, sf, d, toLamp, lp.w, S.lamp_dist[l]);
        }
        if (MODE == HIPEXEC_MODE_CPU && spawn && secondary_ray(S, sf.mat, sf.kr, d, sf.N, w, bd)) {
            bounce = true;
            bo = sf.P;
        }
    }
    return make_float4(ret.x, ret.y, ret.z, w);
}

// ---------------------------------------------------------------------------------------
// k_shade: one thread per ray of the current level.  Writes local[s] = shade_sample(...), clears the
// occlusion bits of s, and appends the bounce ray (ray_tracer.cpp:190-196) to next_rays unless
// next_rays is null (then k_bounce has queued the bounce rays of this level already).
// ---------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void k_shade(DScene S, const float4 *__restrict__ rays, const float4 *__restrict__ hits,
                                               uint32_t n, uint32_t *__restrict__ occl, uint32_t lw,
                                               float4 *__restrict__ local, bool spawn,
                                               float4 *__restrict__ next_rays, uint32_t *__restrict__ next_count,
                                               const uint32_t *__restrict__ n_dev)
{
    n = queue_fill(n, n_dev);
    uint32_t j = blockIdx.x * 256u + threadIdx.x;
    bool valid = j < n;
    f3 o, d; uint32_t s, aux;
    load_ray(rays, valid ? j : 0u, o, d, s, aux);
    float4 h = hits[valid ? j : 0u];
    bool bounce = false;
    f3 bo = mk(0.f, 0.f, 0.f), bd = mk(0.f, 0.f, 1.f);
    if (valid && __float_as_int(h.w) >= -1) {
        local[s] = shade_sample<MODE>(S, o, d, h, occl + (size_t)s * lw, spawn, bounce, bo, bd);
        for (uint32_t k = 0; k < lw; k++) occl[(size_t)s * lw + k] = 0u;
    }
    if (spawn && next_rays) {
        __shared__ uint32_t append_scratch[5];
        uint32_t pos = block_append(bounce, next_count, append_scratch);
        if (bounce) store_ray(next_rays, pos, bo, bd, s, 0u);
    }
}

// ---------------------------------------------------------------------------------------
// k_bounce: the bounce rays of a level (ray_tracer.cpp:190-196) straight from its hits, before
// the level is lit, so that the next level can be traced while this one's shadow rays and
// shading run beside it.  The same expressions as in k_shade, hence the same rays.
// ---------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void k_bounce(DScene S, const float4 *__restrict__ rays, const float4 *__restrict__ hits,
                                                uint32_t n, float4 *__restrict__ next_rays, uint32_t *__restrict__ next_count,
                                                const uint32_t *__restrict__ n_dev)
{
    n = queue_fill(n, n_dev);
    uint32_t j = blockIdx.x * 256u + threadIdx.x;
    bool valid = j < n;
    f3 o, d; uint32_t s, aux;
    load_ray(rays, valid ? j : 0u, o, d, s, aux);
    float4 h = hits[valid ? j : 0u];
    bool bounce = false;
    f3 bo = mk(0.f, 0.f, 0.f), bd = mk(0.f, 0.f, 1.f);
    if (valid && __float_as_int(h.w) >= 0) {
        Surf sf = surface(S, o, d, h, false);
        float kr = S.mat_b[sf.mat].z, w;
        if (MODE == HIPEXEC_MODE_CPU && secondary_ray(S, sf.mat, kr, d, sf.N, w, bd)) {
            bounce = true;
            bo = sf.P;
        }
    }
    __shared__ uint32_t append_scratch[5];
    uint32_t pos = block_append(bounce, next_count, append_scratch);
    if (bounce) store_ray(next_rays, pos, bo, bd, s, 0u);
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
        uint32_t s = sample_index<SPP>(F, slot, r);
        f3 t = fold_levels(local, level_stride, n_levels, s);
        c = (r == 0) ? t : (c + t);
        if (out_ids) out_ids[((size_t)lrow * F.out_w + (x - F.cx0)) * SPP + r] = __float_as_int(hits0[s].w);
    }
    if (MODE == HIPEXEC_MODE_CPU) c = c / 4.0f;
    if (tone) c = mk(pow_cr(c.x / 2.2f, 0.3f), pow_cr(c.y / 2.2f, 0.3f), pow_cr(c.z / 2.2f, 0.3f));
    out_rgba[(size_t)lrow * F.out_w + (x - F.cx0)] = make_float4(c.x, c.y, c.z, 1.0f);
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
