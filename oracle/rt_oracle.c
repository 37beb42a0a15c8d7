/*
 * rt_oracle.c — CPU restatement of the reference tracer.  TEST INFRASTRUCTURE ONLY; see
 * rt_oracle.h for the rules and for why parity is unpinned.
 *
 * Every function cites the reference lines it follows.  All arithmetic is binary32 in the
 * reference's source order; build with -O2 -ffp-contract=off and no -march so that, like the
 * reference's own build (CMakeLists.txt:15-16), no FMA is ever formed.  Comparisons against
 * EPS / 0.0 / 1.0 are made in double exactly as the sources do (config.h:20 is a double
 * literal).
 */
#include "rt_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define EPS (0.0001)          /* config.h:20 */
#define SUB_BLOCK 48          /* config.h:18-19 */

typedef struct { float x, y, z; } v3;

/* ---- glm primitives (canonical scalar formulas) ------------------------------------- */
static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 ld3(const float *p) { return V(p[0], p[1], p[2]); }
static inline void st3(float *p, v3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
static inline v3 add(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mulv(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 muls(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 smul(float s, v3 a) { return V(s * a.x, s * a.y, s * a.z); }
static inline v3 divs(v3 a, float s) { return V(a.x / s, a.y / s, a.z / s); }
static inline v3 neg(v3 a) { return V(-a.x, -a.y, -a.z); }
static inline float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 cross(v3 x, v3 y) {
    return V(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
static inline v3 normalize(v3 v) { return muls(v, 1.0f / sqrtf(dot(v, v))); }
/* row vector * mat3 with column i of the matrix = JSON transform row i (scene.cpp:125-132) */
static inline v3 vecmat(v3 v, const float *m) {
    return V(m[0] * v.x + m[1] * v.y + m[2] * v.z,
             m[3] * v.x + m[4] * v.y + m[5] * v.z,
             m[6] * v.x + m[7] * v.y + m[8] * v.z);
}

/* ---- ray / triangle ------------------------------------------------------------------- */
/* ray_tracer.cpp:300-329 */
static int tri_hit_cpu(v3 p, v3 e1, v3 e2, v3 o, v3 d, float *t, float *u, float *v, v3 *pt)
{
    v3 q = sub(o, p);
    float det = dot(d, cross(e2, e1));
    if ((double)fabsf(det) < EPS) return 0;
    float dett = dot(q, cross(e1, e2));
    float tt = dett / det;
    if ((double)tt < EPS) return 0;
    float detu = dot(d, cross(e2, q));
    float detv = dot(d, cross(q, e1));
    float uu = detu / det;
    float vv = detv / det;
    if ((double)uu < 0.0 || (double)vv < 0.0 || (double)(vv + uu) > 1.0) return 0;
    *t = tt; *u = uu; *v = vv;
    *pt = add(o, smul(tt, d));
    return 1;
}

/* cl_kernels.c:106-147 (t,u,v,point are always produced, like the kernel's unconditional stores) */
static int tri_hit_cl(v3 p, v3 e1, v3 e2, v3 o, v3 d, float *t, float *u, float *v, v3 *pt)
{
    v3 q = sub(o, p);
    v3 c21 = cross(e2, e1);
    v3 c12 = neg(c21);
    v3 ceq = cross(e2, q);
    v3 cqe = cross(q, e1);
    float det = dot(d, c21);
    float dett = dot(q, c12);
    float detu = dot(d, ceq);
    float detv = dot(d, cqe);
    float tt = dett / det, uu = detu / det, vv = detv / det;
    *t = tt; *u = uu; *v = vv;
    *pt = add(o, smul(tt, d));
    int hit = (det > 0.001f || det < -0.001f);
    hit = hit && (tt > 0.001f);
    hit = hit && (uu > 0.0f && vv > 0.0f && uu + vv < 1.0f);
    return hit;
}

int orc_triangle_hit(int mode, const float p[3], const float e1[3], const float e2[3],
                     const float o[3], const float d[3], float tuv[3], float point[3])
{
    float t = 0, u = 0, v = 0; v3 pt = V(0, 0, 0);
    int hit = (mode == HIPEXEC_MODE_CL)
        ? tri_hit_cl(ld3(p), ld3(e1), ld3(e2), ld3(o), ld3(d), &t, &u, &v, &pt)
        : tri_hit_cpu(ld3(p), ld3(e1), ld3(e2), ld3(o), ld3(d), &t, &u, &v, &pt);
    tuv[0] = t; tuv[1] = u; tuv[2] = v;
    if (point) st3(point, pt);
    return hit;
}

/* ---- ray / sphere --------------------------------------------------------------------- */
/* ray_tracer.cpp:332-373 */
static int sph_hit_cpu(v3 c, float r, v3 o, v3 d, float *t, v3 *pt, v3 *nrm)
{
    v3 v = sub(o, c);
    float a = dot(d, d);
    float b = 2 * dot(v, d);
    float cc = dot(v, v) - r * r;
    float D = b * b - 4 * a * cc;
    float tt;
    if (D < 0) {
        return 0;
    } else if ((double)fabsf(D) < EPS) {
        tt = -b / 2 * a;
        if (!((double)tt > EPS)) return 0;
    } else {
        float s = sqrtf(D);
        float t1 = (-b + s) / (2 * a);
        float t2 = (-b - s) / (2 * a);
        if ((double)t1 > EPS && (double)t2 > EPS) tt = t1 < t2 ? t1 : t2;
        else if ((double)t1 > EPS) tt = t1;
        else if ((double)t2 > EPS) tt = t2;
        else return 0;
    }
    *t = tt;
    *pt = add(o, smul(tt, d));
    *nrm = divs(sub(*pt, c), r);
    return 1;
}

/* cl_kernels.c:197-236 */
static int sph_hit_cl(v3 c, float r, v3 o, v3 d, float *t, v3 *pt, v3 *nrm)
{
    v3 v = sub(o, c);
    float a = dot(d, d);
    float b = 2 * dot(v, d);
    float cc = dot(v, v) - (r * r);
    float D = (b * b) - (4 * a * cc);
    float tt = 0.0f;
    if (D < 0.0f) {
        return 0;
    } else if (D < 0.0001f && D > -0.000f) {
        tt = -b / 2.0f * a;
        if (tt < 0.0f) return 0;
    } else {
        float s = sqrtf(D);
        float t1 = (-b + s) / (2.0f * a);
        float t2 = (-b - s) / (2.0f * a);
        if (t1 > 0.0001f && t2 > 0.0001f) tt = t1 < t2 ? t1 : t2;
        else if (t1 > 0.0001f) tt = t1;
        else if (t2 > 0.0001f) tt = t2;
        else return 0;
    }
    *t = tt;
    *pt = add(o, smul(tt, d));
    *nrm = divs(sub(*pt, c), r);
    return 1;
}

int orc_sphere_hit(int mode, const float c[3], float r, const float o[3], const float d[3],
                   float *t, float point[3], float norm[3])
{
    float tt = 0; v3 pt = V(0, 0, 0), n = V(0, 0, 0);
    int hit = (mode == HIPEXEC_MODE_CL) ? sph_hit_cl(ld3(c), r, ld3(o), ld3(d), &tt, &pt, &n)
                                        : sph_hit_cpu(ld3(c), r, ld3(o), ld3(d), &tt, &pt, &n);
    *t = tt;
    if (point) st3(point, pt);
    if (norm) st3(norm, n);
    return hit;
}

/* ---- texture -------------------------------------------------------------------------- */
/* tex_image.h:34-44.  The index expression is reproduced literally (wrap test of y against
 * the width, row stride taken from the height, 32-bit unsigned products).  An index outside
 * the image is undefined behaviour in the reference; here it yields transparent black. */
void orc_tex_get(const hipexec_scene_desc *s, int tex, unsigned x, unsigned y, float rgba[4])
{
    unsigned W = s->tex_w[tex], H = s->tex_h[tex];
    const uint8_t *img = s->tex_rgba + s->tex_offset[tex];
    uint64_t size = (uint64_t)W * H * 4;
    unsigned realX = x >= W ? x % W : x;
    unsigned realY = y >= W ? y % H : y;
    unsigned index = (unsigned)(size - (uint64_t)(unsigned)((realY + 1) * H * 4) + (uint64_t)(unsigned)(realX * 4));
    if ((uint64_t)index + 3 >= size) { rgba[0] = rgba[1] = rgba[2] = rgba[3] = 0.0f; return; }
    rgba[0] = (float)img[index] / 255.0f;
    rgba[1] = (float)img[index + 1] / 255.0f;
    rgba[2] = (float)img[index + 2] / 255.0f;
    rgba[3] = (float)img[index + 3] / 255.0f;
}

static unsigned f2u_x86(float f);

/* Optional, not in the reference: the four texels around (ux - 0.5, uy - 0.5), each fetched with
 * orc_tex_get (same wrap and flip), blended in x then in y.  Texel i is centred at i + 0.5, so
 * at texel centres this equals the reference's lookup. */
void orc_tex_bilinear(const hipexec_scene_desc *s, int tex, float ux, float uy, float rgba[4])
{
    float fx = ux - 0.5f, fy = uy - 0.5f;
    float x0 = floorf(fx), y0 = floorf(fy);
    float ax = fx - x0, ay = fy - y0;
    float c00[4], c10[4], c01[4], c11[4];
    orc_tex_get(s, tex, f2u_x86(x0), f2u_x86(y0), c00);
    orc_tex_get(s, tex, f2u_x86(x0 + 1.0f), f2u_x86(y0), c10);
    orc_tex_get(s, tex, f2u_x86(x0), f2u_x86(y0 + 1.0f), c01);
    orc_tex_get(s, tex, f2u_x86(x0 + 1.0f), f2u_x86(y0 + 1.0f), c11);
    for (int k = 0; k < 4; k++) {
        float top = c00[k] * (1.0f - ax) + c10[k] * ax;
        float bot = c01[k] * (1.0f - ax) + c11[k] * ax;
        rgba[k] = top * (1.0f - ay) + bot * ay;
    }
}

static __thread int tl_bilinear = 0;      /* set per worker thread from hipexec_render_cfg.flags */
static __thread int tl_refract = 0;       /* likewise: HIPEXEC_FLAG_REFRACT */
static __thread int tl_ao = 0;            /* likewise: HIPEXEC_FLAG_AO */

/* (unsigned int) of a float as x86-64 gcc evaluates it: truncate to a 64-bit integer, keep
 * the low 32 bits (ray_tracer.cpp:236-237).  Out-of-range and NaN give 0 like cvttss2si. */
static unsigned f2u_x86(float f)
{
    if (!(f > -9.2233720368547758e18f && f < 9.2233720368547758e18f)) return 0u;
    return (unsigned)(uint64_t)(int64_t)f;
}

/* ---- materials ------------------------------------------------------------------------ */
typedef struct { v3 color; float kd, ks, hard, kr, kt, ior; int tex; } mat_t;

static mat_t get_mat(const hipexec_scene_desc *s, int m)
{
    mat_t r; r.color = V(0, 0, 0); r.kd = r.ks = r.hard = r.kr = r.kt = r.ior = 0.0f; r.tex = -1;   /* scene.h:32-44 */
    if (m >= 0 && (uint32_t)m < s->n_mat) {
        r.color = ld3(s->mat_color + 3 * m);
        r.kd = s->mat_kd[m]; r.ks = s->mat_ks[m]; r.hard = s->mat_hard[m]; r.kr = s->mat_kr[m];
        r.tex = s->mat_tex ? s->mat_tex[m] : -1;
        r.kt = s->mat_kt ? s->mat_kt[m] : 0.0f;              /* scene.h:28-29: declared, never loaded by the reference */
        r.ior = s->mat_ior ? s->mat_ior[m] : 0.0f;
    }
    return r;
}

typedef struct {
    int isHit, id, mat;
    float t, u, v;
    v3 point, norm, color;
} hit_t;

/* colour of a triangle hit: ray_tracer.cpp:232-258 == ray_tracer_cl.cpp:216-224 */
static v3 tri_color(const hipexec_scene_desc *s, uint32_t i, const mat_t *m, float u, float v)
{
    if (m->tex >= 0 && s->tri_uv) {
        const float *uv = s->tri_uv + 6 * (size_t)i;
        float ux = uv[0] + uv[2] * u + uv[4] * v;
        float uy = uv[1] + uv[3] * u + uv[5] * v;
        float rgba[4];
        if (tl_bilinear) orc_tex_bilinear(s, m->tex, ux, uy, rgba);
        else orc_tex_get(s, m->tex, f2u_x86(ux), f2u_x86(uy), rgba);
        v3 texc = smul(rgba[3], V(rgba[0], rgba[1], rgba[2]));
        return add(texc, smul(1 - rgba[3], m->color));
    }
    return m->color;
}

/* ---- closest hit ---------------------------------------------------------------------- */
static hit_t closest_hit(const hipexec_scene_desc *s, int mode, v3 o, v3 d)
{
    int cl = (mode == HIPEXEC_MODE_CL);
    int tHit = 0; uint32_t tIdx = 0; float tT = 0, tU = 0, tV = 0; v3 tP = V(0, 0, 0);
    for (uint32_t i = 0; i < s->n_tri; i++) {
        float t, u, v; v3 pt;
        v3 p = ld3(s->tri_p + 3 * (size_t)i), e1 = ld3(s->tri_e1 + 3 * (size_t)i), e2 = ld3(s->tri_e2 + 3 * (size_t)i);
        int h = cl ? tri_hit_cl(p, e1, e2, o, d, &t, &u, &v, &pt) : tri_hit_cpu(p, e1, e2, o, d, &t, &u, &v, &pt);
        /* ray_tracer.cpp:214 (hit.t > 0 required) vs opencl_executor.cpp:242 (not required) */
        if (h && (cl || t > 0) && (!tHit || t < tT)) { tHit = 1; tIdx = i; tT = t; tU = u; tV = v; tP = pt; }
    }
    int sHit = 0; uint32_t sIdx = 0; float sT = 0; v3 sP = V(0, 0, 0), sN = V(0, 0, 0);
    for (uint32_t i = 0; i < s->n_sph; i++) {
        float t; v3 pt, n;
        v3 c = ld3(s->sph_center + 3 * (size_t)i);
        int h = cl ? sph_hit_cl(c, s->sph_radius[i], o, d, &t, &pt, &n) : sph_hit_cpu(c, s->sph_radius[i], o, d, &t, &pt, &n);
        if (h && (cl || t > 0) && (!sHit || t < sT)) { sHit = 1; sIdx = i; sT = t; sP = pt; sN = n; }
    }
    int useTri, useSph;
    if (cl) {                       /* ray_tracer_cl.cpp:205-211: triangle wins ties */
        if (tHit) { useSph = sHit && sT < tT; useTri = !useSph; }
        else { useSph = sHit; useTri = 0; }
    } else {                        /* ray_tracer.cpp:230,259: sphere wins ties */
        useTri = tHit && (!sHit || tT < sT);
        useSph = !useTri && sHit;
    }
    hit_t r; memset(&r, 0, sizeof r); r.id = -1; r.mat = -1;
    if (useTri) {
        int mi = s->tri_mat ? s->tri_mat[tIdx] : -1;
        mat_t m = get_mat(s, mi);
        r.isHit = 1; r.id = (int)tIdx; r.mat = (mi >= 0 && (uint32_t)mi < s->n_mat) ? mi : -1;
        r.t = tT; r.u = tU; r.v = tV; r.point = tP; r.norm = ld3(s->tri_norm + 3 * (size_t)tIdx);
        r.color = tri_color(s, tIdx, &m, tU, tV);
    } else if (useSph) {
        int mi = s->sph_mat ? s->sph_mat[sIdx] : -1;
        mat_t m = get_mat(s, mi);
        r.isHit = 1; r.id = (int)(s->n_tri + sIdx); r.mat = (mi >= 0 && (uint32_t)mi < s->n_mat) ? mi : -1;
        r.t = sT; r.point = sP; r.norm = sN; r.color = m.color;
    }
    return r;
}

void orc_closest_hits(const hipexec_scene_desc *s, int mode, const float *rays6, uint32_t n,
                      hipexec_hit *out)
{
    for (uint32_t i = 0; i < n; i++) {
        hit_t h = closest_hit(s, mode, ld3(rays6 + 6 * (size_t)i), ld3(rays6 + 6 * (size_t)i + 3));
        out[i].id = h.id; out[i].mat = h.mat; out[i].t = h.t; out[i].u = h.u; out[i].v = h.v;
        st3(out[i].point, h.point); st3(out[i].norm, h.norm); st3(out[i].color, h.color);
    }
}

/* ---- any hit -------------------------------------------------------------------------- */
/* ray_tracer.cpp:276-297; in CL mode the OR of the two per-kind reductions
 * (ray_tracer_cl.cpp:252-261), which is the same boolean. */
static int any_hit(const hipexec_scene_desc *s, int mode, v3 o, v3 d, hipexec_stats *st)
{
    int cl = (mode == HIPEXEC_MODE_CL);
    if (st) st->rays_shadow++;
    for (uint32_t i = 0; i < s->n_tri; i++) {
        float t, u, v; v3 pt;
        v3 p = ld3(s->tri_p + 3 * (size_t)i), e1 = ld3(s->tri_e1 + 3 * (size_t)i), e2 = ld3(s->tri_e2 + 3 * (size_t)i);
        int h = cl ? tri_hit_cl(p, e1, e2, o, d, &t, &u, &v, &pt) : tri_hit_cpu(p, e1, e2, o, d, &t, &u, &v, &pt);
        if (h) { if (st) st->tests_tri_any += i + 1; return 1; }
    }
    if (st) st->tests_tri_any += s->n_tri;
    for (uint32_t i = 0; i < s->n_sph; i++) {
        float t; v3 pt, n;
        v3 c = ld3(s->sph_center + 3 * (size_t)i);
        int h = cl ? sph_hit_cl(c, s->sph_radius[i], o, d, &t, &pt, &n) : sph_hit_cpu(c, s->sph_radius[i], o, d, &t, &pt, &n);
        if (h) { if (st) st->tests_sph_any += i + 1; return 1; }
    }
    if (st) st->tests_sph_any += s->n_sph;
    return 0;
}

void orc_any_hits(const hipexec_scene_desc *s, int mode, const float *rays6, uint32_t n,
                  char *hits, hipexec_stats *stats)
{
    for (uint32_t i = 0; i < n; i++) {
        int h = any_hit(s, mode, ld3(rays6 + 6 * (size_t)i), ld3(rays6 + 6 * (size_t)i + 3), stats);
        hits[i] = hits[i] || h;
    }
}

/* ---- shading -------------------------------------------------------------------------- */
static void count_closest(const hipexec_scene_desc *s, hipexec_stats *st, int depth)
{
    if (!st) return;
    if (depth == 0) { st->rays_primary++; st->tests_tri_primary += s->n_tri; } else st->rays_bounce++;
    st->tests_tri_closest += s->n_tri;
    st->tests_sph_closest += s->n_sph;
}

static int any_hit(const hipexec_scene_desc *s, int mode, v3 o, v3 d, hipexec_stats *st);

/* Counter-based stand-in for rand() / RAND_MAX of generateRandomRayInHalfSphere (ray_tracer.cpp:426-434): the c-th
 * uniform value in [0, 1) for the hit point P (definition in include/hip_executor.h, HIPEXEC_FLAG_AO). */
static uint32_t pcg(uint32_t v)
{
    v = v * 747796405u + 2891336453u;
    uint32_t w = ((v >> ((v >> 28) + 4u)) ^ v) * 277803737u;
    return (w >> 22) ^ w;
}
static float ao_uniform(v3 P, uint32_t c)
{
    uint32_t bx, by, bz;
    memcpy(&bx, &P.x, 4); memcpy(&by, &P.y, 4); memcpy(&bz, &P.z, 4);
    uint32_t h = pcg(pcg(pcg(pcg(c) ^ bx) ^ by) ^ bz);
    return (float)(h >> 8) * (1.0f / 16777216.0f);
}

/* computeAmbientOcclusion, ray_tracer.cpp:406-424, with the generator above */
static float ambient_occlusion(const hipexec_scene_desc *s, v3 P, v3 N, v3 d, hipexec_stats *st)
{
    float ret = 0.0f;
    for (uint32_t i = 0; i < HIPEXEC_AO_RAYS; i++) {
        v3 realNorm = dot(d, N) < 0.0f ? N : neg(N);
        float rx = fabsf(2.0f * ao_uniform(P, 3 * i)) - 1.0f;
        float ry = fabsf(2.0f * ao_uniform(P, 3 * i + 1)) - 1.0f;
        float rz = fabsf(2.0f * ao_uniform(P, 3 * i + 2)) - 1.0f;
        v3 r = V(rx, ry, rz);
        if ((double)dot(r, realNorm) < EPS) r = neg(r);
        if (!any_hit(s, HIPEXEC_MODE_CPU, P, r, st)) ret += s->ambient_factor;
    }
    return ret / (float)HIPEXEC_AO_RAYS;
}

/* traceRay, ray_tracer.cpp:156-201.  max_depth = 0 stands for a build without ENABLE_REFLECTION. */
static v3 trace_cpu(const hipexec_scene_desc *s, v3 o, v3 d, int depth, int max_depth,
                    hipexec_stats *st, int *id_out)
{
    count_closest(s, st, depth);
    hit_t hit = closest_hit(s, HIPEXEC_MODE_CPU, o, d);
    if (id_out) *id_out = hit.id;
    if (!hit.isHit) return ld3(s->horizon);
    mat_t m = get_mat(s, hit.mat);
    v3 ret = ld3(s->ambient);
    if (tl_ao) ret = add(ret, muls(hit.color, ambient_occlusion(s, hit.point, hit.norm, d, st)));      /* ray_tracer.cpp:167-169 */
    for (uint32_t l = 0; l < s->n_lamp; l++) {
        int shaded = 0;
        v3 toLamp = sub(ld3(s->lamp_pos + 3 * l), hit.point);
        float dotWithLamp = dot(toLamp, hit.norm);
        float dotWithDir = dot(d, hit.norm);
        if ((dotWithDir < 0.0 && dotWithLamp < 0.0) || (dotWithDir > 0.0 && dotWithLamp > 0.0)) shaded = 1;
        if (!shaded) shaded |= any_hit(s, HIPEXEC_MODE_CPU, hit.point, toLamp, st);
        if (!shaded) {
            /* computeDiffusiveLight, ray_tracer.cpp:376-385 */
            float dd = fabsf(dot(hit.norm, toLamp));
            float sqrLength = dot(toLamp, toLamp);
            float diff = s->lamp_intensity[l] * s->lamp_distance[l] * dd / sqrLength;
            ret = add(ret, muls(muls(hit.color, m.kd), diff));
            /* computePhongLight, ray_tracer.cpp:388-403 */
            v3 refl = normalize(sub(toLamp, muls(smul(2.0f, hit.norm), dot(toLamp, hit.norm))));
            float sd = dot(refl, d);
            float phong = 0.0f;
            if (sd > 0.0f) phong = (m.hard * s->lamp_distance[l] / dot(toLamp, toLamp)) * powf(sd, m.hard);
            ret = add(ret, muls(muls(hit.color, m.ks), phong));
        }
    }
    if (tl_refract && max_depth > 0 && (double)m.kt > EPS && depth < max_depth) {
        /* HIPEXEC_FLAG_REFRACT (NOT reference behaviour; definition in include/hip_executor.h): the secondary ray of a
         * transmissive material is the refracted one, or the mirrored one under total internal reflection. */
        float n = m.ior > 0.0f ? m.ior : 1.0f;
        float dl = sqrtf(dot(d, d));
        v3 i = muls(d, 1.0f / dl);
        float c = dot(i, hit.norm);
        int entering = c < 0.0f;
        v3 nf = entering ? hit.norm : neg(hit.norm);
        float eta = entering ? 1.0f / n : n;
        float cosi = entering ? -c : c;
        float k = 1.0f - (eta * eta) * (1.0f - cosi * cosi);
        v3 rd;
        if (k < 0.0f) rd = sub(d, muls(smul(2.0f, hit.norm), dot(d, hit.norm)));
        else rd = add(muls(i, eta), muls(nf, eta * cosi - sqrtf(k)));
        ret = add(ret, muls(trace_cpu(s, hit.point, rd, depth + 1, max_depth, st, NULL), m.kt));
    } else if (max_depth > 0 && (double)m.kr > EPS && depth < max_depth) {
        v3 rd = sub(d, muls(smul(2.0f, hit.norm), dot(d, hit.norm)));
        ret = add(ret, muls(trace_cpu(s, hit.point, rd, depth + 1, max_depth, st, NULL), m.kr));
    }
    return ret;
}

/* traceRaysCl for one ray, ray_tracer_cl.cpp:108-178 (the batch is independent per ray and
 * accumulates lamps in scene order, so per-ray evaluation gives the same sums). */
static v3 trace_cl(const hipexec_scene_desc *s, v3 o, v3 d, hipexec_stats *st, int *id_out)
{
    count_closest(s, st, 0);
    hit_t hit = closest_hit(s, HIPEXEC_MODE_CL, o, d);
    if (id_out) *id_out = hit.id;
    v3 col = V(0.0f, 0.0f, 0.0f);
    if (!hit.isHit) return add(col, ld3(s->horizon));
    mat_t m = get_mat(s, hit.mat);
    for (uint32_t l = 0; l < s->n_lamp; l++) {
        v3 toLamp = sub(ld3(s->lamp_pos + 3 * l), hit.point);
        float dotWithLamp = dot(toLamp, hit.norm);
        float dotWithDir = d.x * hit.norm.x + d.y * hit.norm.y + d.z * hit.norm.z;
        if (!((dotWithDir < 0.0 && dotWithLamp >= 0.0) || (dotWithDir > 0.0 && dotWithLamp <= 0.0))) continue;
        if (any_hit(s, HIPEXEC_MODE_CL, hit.point, toLamp, st)) continue;
        float dd = fabsf(dot(hit.norm, toLamp));
        float sqrLength = dot(toLamp, toLamp);
        /* hits.color * kd * intensity * distance * dot / sqrLength, left to right on vec3 */
        v3 c = divs(muls(muls(muls(muls(hit.color, m.kd), s->lamp_intensity[l]), s->lamp_distance[l]), dd), sqrLength);
        col = add(col, c);
        v3 refl = normalize(sub(toLamp, muls(smul(2.0f, hit.norm), dot(toLamp, hit.norm))));
        float sd = dot(refl, d);
        float k = (m.hard * s->lamp_distance[l] / dot(toLamp, toLamp)) * powf(sd, m.hard);
        float spec = (0.0f < k) ? k : 0.0f;      /* std::max(0.0f, k): NaN -> 0 */
        col = add(col, muls(muls(hit.color, m.ks), spec));
    }
    return col;
}

void orc_trace_rays(const hipexec_scene_desc *s, int mode, int max_depth, const float *rays6,
                    uint32_t n, float *colours3, hipexec_stats *stats)
{
    for (uint32_t i = 0; i < n; i++) {
        v3 o = ld3(rays6 + 6 * (size_t)i), d = ld3(rays6 + 6 * (size_t)i + 3);
        v3 c = (mode == HIPEXEC_MODE_CL) ? trace_cl(s, o, d, stats, NULL)
                                         : trace_cpu(s, o, d, 0, max_depth, stats, NULL);
        float *dst = colours3 + 3 * (size_t)i;
        if (mode == HIPEXEC_MODE_CL) {
            st3(dst, add(ld3(dst), c));
        } else {
            st3(dst, add(ld3(dst), c));
        }
    }
}

/* ---- frame ---------------------------------------------------------------------------- */
typedef struct {
    const hipexec_scene_desc *s; const hipexec_render_cfg *cfg;
    int x, y, w, h;                 /* region in the full frame                         */
    float *rgb, *lin; int32_t *ids; /* region-sized outputs                             */
} job_t;

static float tone1(float c) { return powf(c / 2.2f, 0.3f); }

/* pixel loop of renderScene (ray_tracer.cpp:25-53) / renderSubBlockCl (ray_tracer_cl.cpp:69-104)
 * over the sub-rectangle [bx,bx+bw) x [by,by+bh) of the region */
static void render_block(const job_t *J, int bx, int by, int bw, int bh, hipexec_stats *st)
{
    const hipexec_scene_desc *s = J->s;
    int width = J->cfg->width, height = J->cfg->height;
    int cl = (J->cfg->mode == HIPEXEC_MODE_CL);
    tl_bilinear = (J->cfg->flags & HIPEXEC_FLAG_BILINEAR) != 0;
    tl_refract = (J->cfg->flags & HIPEXEC_FLAG_REFRACT) != 0;
    tl_ao = (J->cfg->flags & HIPEXEC_FLAG_AO) != 0;
    int spp = cl ? 1 : 4;
    float camHeight = 0.5;
    float camWidth = (float)width * (camHeight / (float)height);
    float camDist = 1.0;
    float dh = camHeight / (float)height;
    float dw = camWidth / (float)width;
    v3 camPos = ld3(s->cam_pos);
    for (int i = by; i < by + bh; i++) {
        for (int j = bx; j < bx + bw; j++) {
            float rayX = -(camWidth / 2) + (j + J->x) * dw;
            float rayY = -(camHeight / 2) + (i + J->y) * dh;
            v3 rayCamDir = V(rayX, rayY, -camDist);
            v3 rayWorldDir = vecmat(rayCamDir, s->cam_mat);
            v3 c; int id[4] = {-1, -1, -1, -1};
            if (cl) {
                c = trace_cl(s, camPos, normalize(rayWorldDir), st, &id[0]);
            } else {
                v3 rayWorldDx = vecmat(V(dw / 2.0f, 0.0f, 0.0f), s->cam_mat);
                v3 rayWorldDy = vecmat(V(0.0f, dh / 2.0f, 0.0f), s->cam_mat);
                int md = J->cfg->max_depth;
                c = trace_cpu(s, camPos, normalize(rayWorldDir), 0, md, st, &id[0]);
                c = add(c, trace_cpu(s, camPos, normalize(add(rayWorldDir, rayWorldDx)), 0, md, st, &id[1]));
                c = add(c, trace_cpu(s, camPos, normalize(add(rayWorldDir, rayWorldDy)), 0, md, st, &id[2]));
                c = add(c, trace_cpu(s, camPos, normalize(add(add(rayWorldDir, rayWorldDx), rayWorldDy)), 0, md, st, &id[3]));
                c = divs(c, 4.0f);
            }
            size_t px = (size_t)i * J->w + j;
            if (J->lin) st3(J->lin + 3 * px, c);
            if (J->rgb) {
                if (J->cfg->tone) st3(J->rgb + 3 * px, V(tone1(c.x), tone1(c.y), tone1(c.z)));
                else st3(J->rgb + 3 * px, c);
            }
            if (J->ids) for (int r = 0; r < spp; r++) J->ids[px * spp + r] = id[r];
        }
    }
}

typedef struct {
    const job_t *J; int n_blocks; const int *blocks;   /* x,y,w,h relative to the region */
    int next; pthread_mutex_t mu; hipexec_stats st;
} pool_t;

static void stats_add(hipexec_stats *a, const hipexec_stats *b)
{
    a->rays_primary += b->rays_primary; a->rays_bounce += b->rays_bounce; a->rays_shadow += b->rays_shadow;
    a->tests_tri_closest += b->tests_tri_closest; a->tests_sph_closest += b->tests_sph_closest;
    a->tests_tri_any += b->tests_tri_any; a->tests_sph_any += b->tests_sph_any;
    a->tests_tri_primary += b->tests_tri_primary;
}

/* taskProcessor, ray_tracer.cpp:124-153: pop a block under a mutex, render it */
static void *worker(void *arg)
{
    pool_t *P = (pool_t *)arg;
    hipexec_stats local; memset(&local, 0, sizeof local);
    for (;;) {
        pthread_mutex_lock(&P->mu);
        int k = P->next < P->n_blocks ? P->next++ : -1;
        pthread_mutex_unlock(&P->mu);
        if (k < 0) break;
        const int *b = P->blocks + 4 * k;
        render_block(P->J, b[0], b[1], b[2], b[3], &local);
    }
    pthread_mutex_lock(&P->mu);
    stats_add(&P->st, &local);
    pthread_mutex_unlock(&P->mu);
    return NULL;
}

static void run_pool(const job_t *J, int n_blocks, const int *blocks, hipexec_stats *stats, int n_threads)
{
    pool_t P; memset(&P, 0, sizeof P);
    P.J = J; P.n_blocks = n_blocks; P.blocks = blocks;
    pthread_mutex_init(&P.mu, NULL);
    if (n_threads <= 1) {
        worker(&P);
    } else {
        pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
        for (int i = 0; i < n_threads; i++) pthread_create(&th[i], NULL, worker, &P);
        for (int i = 0; i < n_threads; i++) pthread_join(th[i], NULL);
        free(th);
    }
    pthread_mutex_destroy(&P.mu);
    if (stats) stats_add(stats, &P.st);
}

void orc_render_region(const hipexec_scene_desc *s, const hipexec_render_cfg *cfg,
                       int x, int y, int w, int h,
                       float *out_rgb, float *out_linear, int32_t *out_ids,
                       hipexec_stats *stats, int n_threads)
{
    job_t J = {s, cfg, x, y, w, h, out_rgb, out_linear, out_ids};
    /* block list of renderParallel, ray_tracer.cpp:69-113 (48x48 blocks, ragged right/bottom) */
    int nbx = (w + SUB_BLOCK - 1) / SUB_BLOCK, nby = (h + SUB_BLOCK - 1) / SUB_BLOCK;
    int *blocks = (int *)malloc(sizeof(int) * 4 * (size_t)(nbx * nby > 0 ? nbx * nby : 1));
    int n = 0;
    for (int j = 0; j < nby; j++)
        for (int i = 0; i < nbx; i++) {
            int bx = i * SUB_BLOCK, by = j * SUB_BLOCK;
            blocks[4 * n] = bx; blocks[4 * n + 1] = by;
            blocks[4 * n + 2] = w - bx < SUB_BLOCK ? w - bx : SUB_BLOCK;
            blocks[4 * n + 3] = h - by < SUB_BLOCK ? h - by : SUB_BLOCK;
            n++;
        }
    run_pool(&J, n, blocks, stats, n_threads);
    free(blocks);
}

void orc_render_blocks(const hipexec_scene_desc *s, const hipexec_render_cfg *cfg,
                       int n_blocks, const int *blocks_xywh, hipexec_stats *stats, int n_threads)
{
    job_t J = {s, cfg, 0, 0, cfg->width, cfg->height, NULL, NULL, NULL};
    run_pool(&J, n_blocks, blocks_xywh, stats, n_threads);
}
