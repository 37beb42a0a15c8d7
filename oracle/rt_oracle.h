/*
 * rt_oracle.h — CPU restatement of the reference tracer, TEST INFRASTRUCTURE ONLY.
 *
 * This is the parity oracle and the CPU baseline of the build; nothing under oracle/ is
 * part of the product.  Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of
 * bench.py may load it.
 *
 * PARITY UNPINNED: the reference (Vlad8161/RayTracer) ships no tests, golden images or
 * sample scenes, and its sources cannot be compiled in this image because its glm
 * dependency is an empty, unpinned git submodule (.gitmodules:4-6).  The arithmetic below
 * is therefore a restatement from reading the sources, using glm's canonical scalar
 * formulas for dot/cross/normalize/vec*mat, and is checked only against hand-derived
 * known answers and an independent numpy restatement in tests/.
 *
 * The scene is passed in the flattened form of include/hip_executor.h so the oracle and
 * the HIP executor consume byte-identical inputs.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include "../include/hip_executor.h"

#ifdef __cplusplus
extern "C" {
#endif

/* computeTriangleHit (ray_tracer.cpp:300-329) for HIPEXEC_MODE_CPU, kernel triangleHit
 * (cl_kernels.c:90-160) for HIPEXEC_MODE_CL.  tuv receives t,u,v (always written in CL
 * mode, like the kernel does; written on hit in CPU mode).  Returns isHit. */
int orc_triangle_hit(int mode, const float p[3], const float e1[3], const float e2[3],
                     const float o[3], const float d[3], float tuv[3], float point[3]);

/* computeSphereHit (ray_tracer.cpp:332-373) / kernel sphereHit (cl_kernels.c:181-246). */
int orc_sphere_hit(int mode, const float c[3], float r, const float o[3], const float d[3],
                   float *t, float point[3], float norm[3]);

/* tex_image::get (tex_image.h:34-44) on texture `tex` of the scene; rgba receives 4 floats. */
void orc_tex_get(const hipexec_scene_desc *s, int tex, unsigned x, unsigned y, float rgba[4]);
/* optional non-reference mode HIPEXEC_FLAG_BILINEAR (texel units, texel i centred at i + 0.5) */
void orc_tex_bilinear(const hipexec_scene_desc *s, int tex, float ux, float uy, float rgba[4]);

/* computeClosestHit (ray_tracer.cpp:204-273) / computeClosestHitsCl (ray_tracer_cl.cpp:181-242
 * with the host reductions of opencl_executor.cpp:217-259, 406-444). */
void orc_closest_hits(const hipexec_scene_desc *s, int mode, const float *rays6, uint32_t n,
                      hipexec_hit *out);

/* computeAnyHit (ray_tracer.cpp:276-297) / computeAnyHitsCl (ray_tracer_cl.cpp:245-262).
 * hits[i] |= occluded.  stats (optional) accumulates rays_shadow and tests_*_any. */
void orc_any_hits(const hipexec_scene_desc *s, int mode, const float *rays6, uint32_t n,
                  char *hits, hipexec_stats *stats);

/* traceRay (ray_tracer.cpp:156-201) / traceRaysCl (ray_tracer_cl.cpp:108-178): colours3 += . */
void orc_trace_rays(const hipexec_scene_desc *s, int mode, int max_depth, const float *rays6,
                    uint32_t n, float *colours3, hipexec_stats *stats);

/*
 * renderScene(outImg, scene, fullWidth, fullHeight, x, y) (ray_tracer.cpp:15-54) /
 * renderSubBlockCl (ray_tracer_cl.cpp:59-105): renders the w*h region at (x,y) of a
 * cfg->width * cfg->height frame.  Outputs are region-sized, row-major, row 0 = region row 0:
 *   out_rgb    [h][w][3]   toned (cfg->tone=1) or linear pixel colour
 *   out_linear [h][w][3]   optional, colour before the tone curve
 *   out_ids    [h][w][spp] optional, hit ids of the primary samples
 * n_threads > 1 splits the region into 48x48 blocks served by a thread pool the way
 * renderParallel does (ray_tracer.cpp:57-153).  cfg->tile_* are ignored.
 */
void orc_render_region(const hipexec_scene_desc *s, const hipexec_render_cfg *cfg,
                       int x, int y, int w, int h,
                       float *out_rgb, float *out_linear, int32_t *out_ids,
                       hipexec_stats *stats, int n_threads);

/* Traces the pixels of n_blocks rectangles {x,y,w,h} of the full frame on a pool of n_threads
 * workers and discards the colours; only `stats` is produced.  Used to time the CPU baseline
 * on a bounded sample of a large frame. */
void orc_render_blocks(const hipexec_scene_desc *s, const hipexec_render_cfg *cfg,
                       int n_blocks, const int *blocks_xywh, hipexec_stats *stats, int n_threads);

#ifdef __cplusplus
}
#endif
#endif
