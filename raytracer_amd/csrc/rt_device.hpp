// rt_device.hpp — device-side arithmetic of the tracing path (gfx950).
//
// Two kinds of code live here and must not be mixed up:
//   * EXACT code reproduces the reference's binary32 arithmetic operation by operation
//     (ray_tracer.cpp / cl_kernels.c / ray_tracer_cl.cpp, cited per function).  It relies on
//     the translation unit being compiled with -ffp-contract=off and correctly rounded
//     divide/sqrt, and never calls fmaf.
//   * FILTER code decides, conservatively, whether a ray/triangle pair can possibly be a
//     hit.  It uses FMA chains and algebraically rearranged edge functions, is allowed any
//     rounding, and is made sound by a per-pair slack: a pair the exact test accepts is
//     never rejected by the filter.  Every pair that passes the filter is then decided by
//     the exact code, so results do not depend on the filter at all.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "hip_executor.h"

#define DEV __device__ __forceinline__

namespace rt {

struct f3 { float x, y, z; };

DEV f3 mk(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
DEV f3 operator+(f3 a, f3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
DEV f3 operator-(f3 a, f3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
DEV f3 operator-(f3 a) { return mk(-a.x, -a.y, -a.z); }
DEV f3 operator*(f3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
DEV f3 operator*(float s, f3 a) { return mk(s * a.x, s * a.y, s * a.z); }
DEV f3 operator/(f3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }
// glm::dot / cross / normalize, canonical scalar forms
DEV float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
DEV f3 cross(f3 x, f3 y) { return mk(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y); }
DEV f3 normalize(f3 v) { return v * (1.0f / sqrtf(dot(v, v))); }
// row vector * mat3 with matrix column i = cam_mat row i (scene.cpp:125-132)
DEV f3 vecmat(f3 v, const float *m) {
    return mk(m[0] * v.x + m[1] * v.y + m[2] * v.z,
              m[3] * v.x + m[4] * v.y + m[5] * v.z,
              m[6] * v.x + m[7] * v.y + m[8] * v.z);
}
DEV f3 xyz(float4 a) { return mk(a.x, a.y, a.z); }

// Scene in device memory.  Triangle arrays are padded to a multiple of TRI_PAD records with
// records no ray can hit.
constexpr int TRI_PAD = 128;
// The 64-bit work counters of a call exist in CNT_STRIPES copies of CNT_STRIDE words; the host adds the copies up.
constexpr uint32_t CNT_STRIPES = 64, CNT_STRIDE = 32;   // = 2 * LDS_CHUNK triangles: whole LDS chunks of triangle pairs

typedef float v2f __attribute__((ext_vector_type(2)));

// FILTER record of a pair of triangles (2j, 2j+1) for camera rays; every member holds the
// value of triangle 2j in .x and of triangle 2j+1 in .y so that one packed FP32 instruction
// (v_pk_fma_f32 / v_pk_mul_f32) evaluates both.  Edge functions are scaled by 1/slack, so a
// pair "may hit" iff min(u, v, w) >= -theta with theta ~ 1.
struct PairCam {
    v2f ux, uy, uz;      // s * cross(e2, q) / slack
    v2f vx, vy, vz;      // s * cross(q, e1) / slack
    v2f wx, wy, wz;      // s * (c21 - ceq - cqe) / slack
    v2f pad;
};
// A record that must never pass (padding, zero-area triangles) has ux = +1e30, vx = -1e30 and
// zeros elsewhere: min(u,v,w) = -1e30*|d.x| and max(u,v,w) = +1e30*|d.x| fail both tests for
// every direction with |d.x| > 1e-30 (the rest merely take the exact path, which rejects).
// The same record layout, without the sign factor s and with q taken from a lamp instead of
// the camera, serves shadow rays (all shadow rays of a lamp lie on lines through the lamp).
// FILTER record of a pair of triangles for arbitrary rays (see k_setup_gen_pairs)
struct PairGen {
    v2f ud[3], um[3];    // u = dot(ud, d) + dot(um, mo)
    v2f vd[3], vm[3];    // v = dot(vd, d) + dot(vm, mo)
    v2f wd[3], wm[3];    // w = dot(wd, d) + dot(wm, mo)
};
static_assert(sizeof(PairCam) == 80 && sizeof(PairGen) == 144, "record sizes");

struct DScene {
    uint32_t n_tri, n_tri_pad, n_sph, n_lamp, n_mat;
    // EXACT records, 3 float4 per triangle: {p, c21.x} {e1, c21.y} {e2, c21.z}, c21 = cross(e2,e1)
    const float4 *tri_exact;
    // EXACT camera-ray records, 3 float4: {c21, dett} {cross(e2,q), 0} {cross(q,e1), 0}, q = cam_pos - p
    const float4 *tri_primx;
    const PairCam *pair_cam;  // [n_tri_pad / 2]
    const PairGen *pair_gen;  // [n_tri_pad / 2]
    const PairCam *pair_lamp; // [n_lamp][n_tri_pad / 2], edge functions of lines through lamp l
    // Bounding boxes of the index-contiguous chunks of CHUNK_TRIS triangles, 3 float4 per chunk:
    // {centre, A} {half extent, e_abs} {A * |half extent|_1, 0, 0, 0} (see chunk_may_hit)
    const float4 *chunk_box;
    uint32_t cull;           // 1: skip chunks no ray of the wave can hit; 0: scan every chunk
    uint32_t bilinear;       // 1: HIPEXEC_FLAG_BILINEAR for this call (not a reference mode)
    uint32_t refract;        // 1: HIPEXEC_FLAG_REFRACT for this call (not a reference mode)
    uint32_t ao;             // 1: HIPEXEC_FLAG_AO for this call (the reference's ENABLE_AO with a counter-based generator)
    float ambient_factor;    // worldAmbientFactor
    const float2 *mat_c;     // {refractionFactor, refractionHardness = index of refraction}   [n_mat + 1]
    const float *tri_norm;   // [n][3]
    const float *tri_uv;     // [n][6]
    const int *tri_mat;      // [n]  (index n_mat = default material)
    const float4 *sph;       // {centre, radius}
    const int *sph_mat;
    const float4 *mat_a;     // {colour, kd}            [n_mat + 1]
    const float4 *mat_b;     // {ks, hardness, kr, texture index as int bits}
    const uint8_t *tex; const uint64_t *tex_off; const uint32_t *tex_w; const uint32_t *tex_h;
    const float4 *lamp;      // {pos, intensity}
    const float *lamp_dist;
    float cam_pos[3], cam_mat[9], horizon[3], ambient[3];
    float center[3];         // translation used by the FILTER records only
    float rmax;              // radius around `center` the FILTER slack was computed for
    float theta;             // FILTER threshold scale: 1.001 normally, 3e38 disables the filter
};

// ---------------------------------------------------------------------------------------
// EXACT ray/triangle tests
// ---------------------------------------------------------------------------------------
// Float forms of the reference's double comparisons against EPS = 0.0001 (config.h:20):
// 0.0001f is the largest float below 1e-4, so  x < EPS  <=>  x <= 0.0001f  and
// x > EPS  <=>  x > 0.0001f  for every float x.
#define RT_EPSF 0.0001f

// computeTriangleHit, ray_tracer.cpp:300-329
DEV bool tri_exact_cpu(f3 o, f3 d, f3 p, f3 e1, f3 e2, float &t, float &u, float &v) {
    f3 q = o - p;
    float det = dot(d, cross(e2, e1));
    if (fabsf(det) <= RT_EPSF) return false;
    float dett = dot(q, cross(e1, e2));
    float tt = dett / det;
    if (tt <= RT_EPSF) return false;
    float detu = dot(d, cross(e2, q));
    float detv = dot(d, cross(q, e1));
    float uu = detu / det;
    float vv = detv / det;
    if (uu < 0.0f || vv < 0.0f || vv + uu > 1.0f) return false;
    t = tt; u = uu; v = vv;
    return true;
}

// kernel triangleHit, cl_kernels.c:106-147
DEV bool tri_exact_cl(f3 o, f3 d, f3 p, f3 e1, f3 e2, float &t, float &u, float &v) {
    f3 q = o - p;
    f3 c21 = cross(e2, e1);
    f3 c12 = -c21;
    f3 ceq = cross(e2, q);
    f3 cqe = cross(q, e1);
    float det = dot(d, c21);
    float dett = dot(q, c12);
    float detu = dot(d, ceq);
    float detv = dot(d, cqe);
    float tt = dett / det, uu = detu / det, vv = detv / det;
    bool hit = (det > 0.001f || det < -0.001f);
    hit = hit && (tt > 0.001f);
    hit = hit && (uu > 0.0f && vv > 0.0f && uu + vv < 1.0f);
    t = tt; u = uu; v = vv;
    return hit;
}

template <int MODE>
DEV bool tri_exact(f3 o, f3 d, f3 p, f3 e1, f3 e2, float &t, float &u, float &v) {
    if (MODE == HIPEXEC_MODE_CL) return tri_exact_cl(o, d, p, e1, e2, t, u, v);
    return tri_exact_cpu(o, d, p, e1, e2, t, u, v);
}

// The same tests for camera rays, with everything that depends only on (camera, triangle)
// taken from the EXACT camera-ray record.  The record holds the bits the per-ray evaluation
// would produce (q = cam_pos - p is ray independent), so results are unchanged.
template <int MODE>
DEV bool tri_exact_cam(f3 d, f3 c21, float dett, f3 ceq, f3 cqe, float &t, float &u, float &v) {
    float det = dot(d, c21);
    if (MODE == HIPEXEC_MODE_CL) {
        float detu = dot(d, ceq);
        float detv = dot(d, cqe);
        float tt = dett / det, uu = detu / det, vv = detv / det;
        bool hit = (det > 0.001f || det < -0.001f);
        hit = hit && (tt > 0.001f);
        hit = hit && (uu > 0.0f && vv > 0.0f && uu + vv < 1.0f);
        t = tt; u = uu; v = vv;
        return hit;
    }
    if (fabsf(det) <= RT_EPSF) return false;
    float tt = dett / det;
    if (tt <= RT_EPSF) return false;
    float detu = dot(d, ceq);
    float detv = dot(d, cqe);
    float uu = detu / det;
    float vv = detv / det;
    if (uu < 0.0f || vv < 0.0f || vv + uu > 1.0f) return false;
    t = tt; u = uu; v = vv;
    return true;
}

// ---------------------------------------------------------------------------------------
// EXACT ray/sphere tests
// ---------------------------------------------------------------------------------------
// computeSphereHit, ray_tracer.cpp:332-373
DEV bool sph_exact_cpu(f3 o, f3 d, f3 c, float r, float &t) {
    f3 v = o - c;
    float a = dot(d, d);
    float b = 2 * dot(v, d);
    float cc = dot(v, v) - r * r;
    float D = b * b - 4 * a * cc;
    float tt;
    if (D < 0) {
        return false;
    } else if (fabsf(D) <= RT_EPSF) {
        tt = -b / 2 * a;
        if (!(tt > RT_EPSF)) return false;
    } else {
        float s = sqrtf(D);
        float t1 = (-b + s) / (2 * a);
        float t2 = (-b - s) / (2 * a);
        if (t1 > RT_EPSF && t2 > RT_EPSF) tt = t1 < t2 ? t1 : t2;
        else if (t1 > RT_EPSF) tt = t1;
        else if (t2 > RT_EPSF) tt = t2;
        else return false;
    }
    t = tt;
    return true;
}

// kernel sphereHit, cl_kernels.c:197-236
DEV bool sph_exact_cl(f3 o, f3 d, f3 c, float r, float &t) {
    f3 v = o - c;
    float a = dot(d, d);
    float b = 2 * dot(v, d);
    float cc = dot(v, v) - (r * r);
    float D = (b * b) - (4 * a * cc);
    float tt = 0.0f;
    if (D < 0.0f) {
        return false;
    } else if (D < 0.0001f && D > -0.000f) {
        tt = -b / 2.0f * a;
        if (tt < 0.0f) return false;
    } else {
        float s = sqrtf(D);
        float t1 = (-b + s) / (2.0f * a);
        float t2 = (-b - s) / (2.0f * a);
        if (t1 > 0.0001f && t2 > 0.0001f) tt = t1 < t2 ? t1 : t2;
        else if (t1 > 0.0001f) tt = t1;
        else if (t2 > 0.0001f) tt = t2;
        else return false;
    }
    t = tt;
    return true;
}

template <int MODE>
DEV bool sph_exact(f3 o, f3 d, f3 c, float r, float &t) {
    if (MODE == HIPEXEC_MODE_CL) return sph_exact_cl(o, d, c, r, t);
    return sph_exact_cpu(o, d, c, r, t);
}

// ---------------------------------------------------------------------------------------
// Hit records and the triangle/sphere merge
// ---------------------------------------------------------------------------------------
struct Best {        // running closest hit of one ray
    float t, u, v;
    int id;          // -1 none
};
DEV Best best_none() { Best b; b.t = 0.f; b.u = 0.f; b.v = 0.f; b.id = -1; return b; }

// acceptance rule of the linear scans: ray_tracer.cpp:214,224 (needs t > 0) /
// opencl_executor.cpp:242,429 (does not); strict '<' keeps the lowest index on ties.
template <int MODE>
DEV void best_update(Best &b, float t, float u, float v, int id) {
    bool ok = (MODE == HIPEXEC_MODE_CL) ? true : (t > 0);
    if (ok && (b.id < 0 || t < b.t)) { b.t = t; b.u = u; b.v = v; b.id = id; }
}

// all spheres, in index order, for one ray; returns the sphere winner (id = sphere index)
template <int MODE>
DEV Best spheres_closest(const DScene &S, f3 o, f3 d) {
    Best sb = best_none();
    for (uint32_t j = 0; j < S.n_sph; j++) {
        float4 s = S.sph[j];
        float t;
        if (sph_exact<MODE>(o, d, xyz(s), s.w, t)) best_update<MODE>(sb, t, 0.f, 0.f, (int)j);
    }
    return sb;
}

// ray_tracer.cpp:230,259 (sphere wins ties) / ray_tracer_cl.cpp:205-211 (triangle wins ties)
template <int MODE>
DEV float4 merge_hits(const DScene &S, const Best &tb, const Best &sb) {
    bool useTri, useSph;
    if (MODE == HIPEXEC_MODE_CL) {
        if (tb.id >= 0) { useSph = sb.id >= 0 && sb.t < tb.t; useTri = !useSph; }
        else { useSph = sb.id >= 0; useTri = false; }
    } else {
        useTri = tb.id >= 0 && (sb.id < 0 || tb.t < sb.t);
        useSph = !useTri && sb.id >= 0;
    }
    if (useTri) return make_float4(tb.t, tb.u, tb.v, __int_as_float(tb.id));
    if (useSph) return make_float4(sb.t, 0.f, 0.f, __int_as_float((int)S.n_tri + sb.id));
    return make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
}

// ---------------------------------------------------------------------------------------
// Surface at a hit: point, normal, colour, material
// ---------------------------------------------------------------------------------------
struct Surf { f3 P, N, color; float kd, ks, hard, kr; int mat; };

// Secondary ray of a hit on a surface with normal N: the mirrored ray of ray_tracer.cpp:190-196 (weight
// reflectionFactor kr, direction d - 2N(d.N), not re-normalised), or, with HIPEXEC_FLAG_REFRACT on a material whose
// refractionFactor kt exceeds EPS, the refracted ray as include/hip_executor.h defines it (weight kt).  Returns false
// if the path ends here.
DEV bool secondary_ray(const DScene &S, int mat, float kr, f3 d, f3 N, float &w, f3 &bd);

// (unsigned int) of a float the way x86-64 gcc evaluates it (truncate to 64-bit, keep the low
// 32 bits): ray_tracer.cpp:236-237
DEV uint32_t f2u_x86(float f) {
    if (!(f > -9.2233720368547758e18f && f < 9.2233720368547758e18f)) return 0u;
    return (uint32_t)(unsigned long long)(long long)f;
}

// tex_image::get, tex_image.h:34-44 (index expression reproduced literally; an index outside
// the image, undefined in the reference, yields transparent black)
DEV float4 tex_get(const DScene &S, int tex, uint32_t x, uint32_t y) {
    uint32_t W = S.tex_w[tex], H = S.tex_h[tex];
    const uint8_t *img = S.tex + S.tex_off[tex];
    uint64_t size = (uint64_t)W * H * 4;
    uint32_t rx = x >= W ? x % W : x;
    uint32_t ry = y >= W ? y % H : y;
    uint32_t index = (uint32_t)(size - (uint64_t)(uint32_t)((ry + 1) * H * 4) + (uint64_t)(uint32_t)(rx * 4));
    if ((uint64_t)index + 3 >= size) return make_float4(0.f, 0.f, 0.f, 0.f);
    uchar4 px = *reinterpret_cast<const uchar4 *>(img + index);
    return make_float4((float)px.x / 255.0f, (float)px.y / 255.0f, (float)px.z / 255.0f, (float)px.w / 255.0f);
}

// HIPEXEC_FLAG_BILINEAR (not in the reference): the four texels around (ux - 0.5, uy - 0.5), each
// fetched with tex_get (same wrap and flip), blended in x then in y; texel i is centred at i + 0.5.
DEV float4 tex_bilinear(const DScene &S, int tex, float ux, float uy) {
    float fx = ux - 0.5f, fy = uy - 0.5f;
    float x0 = floorf(fx), y0 = floorf(fy);
    float ax = fx - x0, ay = fy - y0;
    float4 c00 = tex_get(S, tex, f2u_x86(x0), f2u_x86(y0)), c10 = tex_get(S, tex, f2u_x86(x0 + 1.0f), f2u_x86(y0));
    float4 c01 = tex_get(S, tex, f2u_x86(x0), f2u_x86(y0 + 1.0f)), c11 = tex_get(S, tex, f2u_x86(x0 + 1.0f), f2u_x86(y0 + 1.0f));
    float a[4] = {c00.x, c00.y, c00.z, c00.w}, b[4] = {c10.x, c10.y, c10.z, c10.w};
    float c[4] = {c01.x, c01.y, c01.z, c01.w}, e[4] = {c11.x, c11.y, c11.z, c11.w}, r[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        float top = a[k] * (1.0f - ax) + b[k] * ax;
        float bot = c[k] * (1.0f - ax) + e[k] * ax;
        r[k] = top * (1.0f - ay) + bot * ay;
    }
    return make_float4(r[0], r[1], r[2], r[3]);
}

// ray_tracer.cpp:232-268 / ray_tracer_cl.cpp:213-240.  `want_color` = false skips the
// material/texture part (shadow-ray generation only needs P and N).
DEV Surf surface(const DScene &S, f3 o, f3 d, float4 h, bool want_color) {
    Surf r;
    int id = __float_as_int(h.w);
    r.P = o + h.x * d;
    int mi;
    if (id < (int)S.n_tri) {
        r.N = mk(S.tri_norm[3 * id], S.tri_norm[3 * id + 1], S.tri_norm[3 * id + 2]);
        mi = S.tri_mat[id];
    } else {
        float4 s = S.sph[id - (int)S.n_tri];
        r.N = (r.P - xyz(s)) / s.w;
        mi = S.sph_mat[id - (int)S.n_tri];
    }
    r.mat = mi;
    r.color = mk(0.f, 0.f, 0.f); r.kd = r.ks = r.hard = r.kr = 0.f;
    if (!want_color) return r;
    float4 ma = S.mat_a[mi], mb = S.mat_b[mi];
    r.color = xyz(ma); r.kd = ma.w; r.ks = mb.x; r.hard = mb.y; r.kr = mb.z;
    int tex = __float_as_int(mb.w);
    if (id < (int)S.n_tri && tex >= 0) {
        const float *uv = S.tri_uv + 6 * (size_t)id;
        float ux = uv[0] + uv[2] * h.y + uv[4] * h.z;
        float uy = uv[1] + uv[3] * h.y + uv[5] * h.z;
        float4 c = S.bilinear ? tex_bilinear(S, tex, ux, uy) : tex_get(S, tex, f2u_x86(ux), f2u_x86(uy));
        r.color = c.w * mk(c.x, c.y, c.z) + (1 - c.w) * xyz(ma);
    }
    return r;
}

DEV bool secondary_ray(const DScene &S, int mat, float kr, f3 d, f3 N, float &w, f3 &bd) {
    if (S.refract) {
        float2 c2 = S.mat_c[mat];
        if (c2.x > 0.0001f) {            // RT_EPSF
            float n = c2.y > 0.0f ? c2.y : 1.0f;
            float dl = sqrtf(dot(d, d));
            f3 i = d * (1.0f / dl);
            float c = dot(i, N);
            bool entering = c < 0.0f;
            f3 nf = entering ? N : -N;
            float eta = entering ? 1.0f / n : n;
            float cosi = entering ? -c : c;
            float k = 1.0f - (eta * eta) * (1.0f - cosi * cosi);
            if (k < 0.0f) bd = d - 2.0f * N * dot(d, N);
            else bd = i * eta + nf * (eta * cosi - sqrtf(k));
            w = c2.x;
            return true;
        }
    }
    if (kr > 0.0001f) {
        w = kr;
        bd = d - 2.0f * N * dot(d, N);
        return true;
    }
    return false;
}

// HIPEXEC_FLAG_AO: the c-th uniform value in [0, 1) for hit point P, and the i-th occlusion ray of
// generateRandomRayInHalfSphere (ray_tracer.cpp:426-434) built from three of them (definition in hip_executor.h)
DEV uint32_t pcg_hash(uint32_t v) {
    v = v * 747796405u + 2891336453u;
    uint32_t w = ((v >> ((v >> 28) + 4u)) ^ v) * 277803737u;
    return (w >> 22) ^ w;
}
DEV float ao_uniform(f3 P, uint32_t c) {
    uint32_t h = pcg_hash(pcg_hash(pcg_hash(pcg_hash(c) ^ __float_as_uint(P.x)) ^ __float_as_uint(P.y)) ^ __float_as_uint(P.z));
    return (float)(h >> 8) * (1.0f / 16777216.0f);
}
DEV f3 ao_ray(f3 P, f3 realNorm, uint32_t i) {
    float rx = fabsf(2.0f * ao_uniform(P, 3u * i)) - 1.0f;
    float ry = fabsf(2.0f * ao_uniform(P, 3u * i + 1u)) - 1.0f;
    float rz = fabsf(2.0f * ao_uniform(P, 3u * i + 2u)) - 1.0f;
    f3 r = mk(rx, ry, rz);
    return (dot(r, realNorm) <= RT_EPSF) ? -r : r;          // `< EPS` in double
}

// powf(x, y) of the shading code.  Evaluated in double and rounded once: correctly rounded
// in all but vanishingly few cases, which is also what glibc's powf (the reference's) gives.
// Whole-number exponents up to 1024 (a specular hardness as Blender exports it is an integer from 1 to 511) take
// square-and-multiply in double: at most 20 multiplications, within 3e-15 relative of the power, so the rounding to
// float differs from the correctly rounded result only where the generic evaluation in double could as well.
DEV bool pow_whole_exponent(float y) { int n = (int)y; return (float)n == y && n >= 0 && n <= 1024; }
DEV float pow_whole(float x, float y) {          // for exponents that pow_whole_exponent accepts
    int n = (int)y;
    double b = (double)x, r = 1.0;
    for (; n; n >>= 1) { if (n & 1) r *= b; b *= b; }
    return (float)r;
}
// The generic evaluation is a function of its own, never inlined: inlined into a lamp loop, its two dozen 64-bit
// polynomial constants are hoisted in front of the loop and, in the kernels that shade after tracing and are short of
// registers, from there into scratch memory (ten 8-byte stores per sample on every path, 0.9 GB of HBM writes per C2
// frame), although hardly a scene has a fractional hardness.  As a call on the rare path it costs the common one nothing:
// k_bvh_shadow_direct 80 VGPRs + 18 spilled -> 76 and none, k_shade 96 -> 76, k_bvh_tail 42 spilled -> 4.
__device__ __attribute__((noinline)) float pow_generic(float x, float y) { return (float)pow((double)x, (double)y); }
DEV float pow_cr(float x, float y) {
    if (pow_whole_exponent(y)) return pow_whole(x, y);
    return pow_generic(x, y);
}

// facing test that decides whether a shadow ray is cast:
// ray_tracer.cpp:174-182 (skip when both dots have the same strict sign) /
// ray_tracer_cl.cpp:138-143 (cast only for opposite, non-strict on the lamp side)
template <int MODE>
DEV bool casts_shadow_ray(float dotWithDir, float dotWithLamp) {
    if (MODE == HIPEXEC_MODE_CL)
        return (dotWithDir < 0.0f && dotWithLamp >= 0.0f) || (dotWithDir > 0.0f && dotWithLamp <= 0.0f);
    return !((dotWithDir < 0.0f && dotWithLamp < 0.0f) || (dotWithDir > 0.0f && dotWithLamp > 0.0f));
}

// contribution of one unoccluded lamp, added to `ret` in the reference's order:
// ray_tracer.cpp:185-187,376-403 / ray_tracer_cl.cpp:159-174
template <int MODE>
DEV f3 add_lamp(f3 ret, const Surf &sf, f3 d, f3 toLamp, float intensity, float distance) {
    float dd = fabsf(dot(sf.N, toLamp));
    float sqrLength = dot(toLamp, toLamp);
    f3 refl = normalize(toLamp - 2.0f * sf.N * dot(toLamp, sf.N));
    float sd = dot(refl, d);
    if (MODE == HIPEXEC_MODE_CL) {
        ret = ret + ((((sf.color * sf.kd) * intensity) * distance) * dd) / sqrLength;
        float k = (sf.hard * distance / dot(toLamp, toLamp)) * pow_cr(sd, sf.hard);
        float spec = (0.0f < k) ? k : 0.0f;
        ret = ret + (sf.color * sf.ks) * spec;
    } else {
        float diff = intensity * distance * dd / sqrLength;
        ret = ret + (sf.color * sf.kd) * diff;
        float phong = 0.0f;
        if (sd > 0.0f) phong = (sf.hard * distance / dot(toLamp, toLamp)) * pow_cr(sd, sf.hard);
        ret = ret + (sf.color * sf.ks) * phong;
    }
    return ret;
}

// ---------------------------------------------------------------------------------------
// wavefront helpers (wave64)
// ---------------------------------------------------------------------------------------
DEV int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// Ballot/popcount compaction: every lane of the wave must call this; lanes with pred get
// consecutive slots of the queue whose fill counter is *counter.  One atomic per wave.
DEV uint32_t wave_append(bool pred, uint32_t *counter) {
    unsigned long long m = __ballot(pred);
    if (m == 0) return 0;
    int leader = __ffsll((long long)m) - 1;
    int lane = lane_id();
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(m));
    base = (uint32_t)__shfl((int)base, leader);
    return base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
}

// The same for a whole 256-thread block: one global atomic per block instead of one per wave
// (atomics on a single address are serialised by the L2; at one per wave they bound k_shade).
// Every thread of the block (64 to 256 threads) must call this; `scratch` is 5 words of LDS.
DEV uint32_t block_append(bool pred, uint32_t *counter, uint32_t *scratch) {
    unsigned long long m = __ballot(pred);
    int lane = lane_id();
    uint32_t w = threadIdx.x >> 6;
    if (lane == 0) scratch[w] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (uint32_t k = 0; k < (blockDim.x >> 6); k++) total += scratch[k];
        scratch[4] = total ? atomicAdd(counter, total) : 0u;
    }
    __syncthreads();
    uint32_t base = scratch[4];
    for (uint32_t k = 0; k < w; k++) base += scratch[k];
    return base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
}

// Sum of a small value over the 64 lanes of a converged wave (the sum must fit 32 bits), the same in every lane:
// inclusive scan inside the rows of 16 lanes (row_shr 1, 2, 4, 8), row totals passed on (row_bcast 15, 31), total read
// from lane 63.  Six DPP additions, no trip through the LDS pipe (a shuffle-based reduction is six ds_bpermute deep).
DEV uint32_t wave_sum_small(uint32_t v) {
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);
    return (uint32_t)__builtin_amdgcn_readlane(x, 63);
}
// the same for any 32-bit value: its halves are summed separately, so nothing can overflow
DEV unsigned long long wave_sum_u32(uint32_t v) {
    return (unsigned long long)wave_sum_small(v & 0xffffu) + ((unsigned long long)wave_sum_small(v >> 16) << 16);
}

DEV unsigned long long wave_sum_u64(unsigned long long v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    return v;   // valid in lane 0
}

}  // namespace rt
