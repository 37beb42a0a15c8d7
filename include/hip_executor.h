/*
 * hip_executor.h — C ABI of the MI355X (gfx950) tracing executor.
 *
 * This library replaces the reference's OpenCL back end (class OpenClExecutor,
 * opencl_executor.h:12-95 / opencl_executor.cpp:8-524) and the device-facing half of its
 * render driver (ray_tracer_cl.cpp:9-262).  Everything here is plain C: opaque handle,
 * pointers, sizes and PODs, so it can be bound from C++ (ray_tracer_hip.cpp), ctypes or
 * any other FFI.  Every function returns 0 on success and a negative HIPEXEC_E_* code on
 * failure and never throws; hipexec_last_error() returns the message for the calling
 * thread's last failure.
 *
 * Threads: different executors may be used from different threads at the same time; the calls
 * on one executor must not overlap (the reference's executor is used the same way, one render
 * call at a time, ray_tracer_cl.cpp:14-55).  An executor may start host threads and streams of
 * its own for a call; they are idle between calls and end with hipexec_destroy().
 *
 * Arithmetic contract: all tracing arithmetic is IEEE-754 binary32, evaluated in the
 * reference's source order with no FMA contraction, correctly rounded divide and sqrt.
 * Two rule sets exist because the reference's two back ends disagree:
 *   HIPEXEC_MODE_CPU  follows ray_tracer.cpp (4 samples per pixel, ambient term, inclusive
 *                     barycentric edges, 1e-4 thresholds, sphere wins ties, optional
 *                     recursive reflection);
 *   HIPEXEC_MODE_CL   follows ray_tracer_cl.cpp + cl_kernels.c (1 sample per pixel, no
 *                     ambient, strict edges, 1e-3 thresholds, triangle wins ties, no
 *                     reflection).
 */
#ifndef HIP_EXECUTOR_H
#define HIP_EXECUTOR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HIPEXEC_MODE_CPU 0
#define HIPEXEC_MODE_CL  1

#define HIPEXEC_OK            0
#define HIPEXEC_E_INVALID    -1   /* bad argument (null pointer, negative size, bad index) */
#define HIPEXEC_E_DEVICE     -2   /* a HIP runtime call failed                               */
#define HIPEXEC_E_NOMEM      -3   /* host or device allocation failed                        */
#define HIPEXEC_E_LIMIT      -4   /* input exceeds a documented limit                        */

/* Rows of a frame are grouped in tiles of HIPEXEC_TILE_ROWS scanlines; a tile is the unit
 * of multi-GPU sharding (tile k of the frame covers rows [8k, 8k+8)). */
#define HIPEXEC_TILE_ROWS 8

typedef struct hipexec hipexec_t;

/*
 * Flattened scene.  Replaces the pointer-chasing `Scene` of scene.h:143-153 at the boundary;
 * values are exactly what scene.cpp:15-135 leaves in the Scene (triangle edges already
 * relative to p, UVs already in texel units, sphere centres and lamp positions already
 * truncated to integers).  All arrays are caller-owned and only read during
 * hipexec_create().  A material index of -1 means "no material" and behaves like the
 * default Material() of scene.h:32-44 (black, all factors 0, untextured).
 */
typedef struct hipexec_scene_desc {
    /* triangles: scene.h:100-120 */
    uint32_t       n_tri;
    const float   *tri_p;      /* [n_tri][3]  first vertex                                  */
    const float   *tri_e1;     /* [n_tri][3]  v1 - v0                                       */
    const float   *tri_e2;     /* [n_tri][3]  v2 - v0                                       */
    const float   *tri_norm;   /* [n_tri][3]  normalize(cross(e1,e2)) as stored by the loader */
    const float   *tri_uv;     /* [n_tri][6]  uvStart.xy, uvU.xy, uvV.xy (texel units); may be NULL */
    const int32_t *tri_mat;    /* [n_tri]     material index or -1                          */
    /* spheres: scene.h:123-131 */
    uint32_t       n_sph;
    const float   *sph_center; /* [n_sph][3]                                                */
    const float   *sph_radius; /* [n_sph]                                                   */
    const int32_t *sph_mat;    /* [n_sph]                                                   */
    /* materials: scene.h:21-45 */
    uint32_t       n_mat;
    const float   *mat_color;  /* [n_mat][3]  diffusiveColor                                */
    const float   *mat_kd;     /* [n_mat]     diffusiveFactor                               */
    const float   *mat_ks;     /* [n_mat]     specularFactor                                */
    const float   *mat_hard;   /* [n_mat]     specularHardness                              */
    const float   *mat_kr;     /* [n_mat]     reflectionFactor                              */
    const int32_t *mat_tex;    /* [n_mat]     texture index or -1 (Material::textured)      */
    /* textures: tex_image.h:13-69, decoded RGBA8 rows top-down as LodePNG leaves them */
    uint32_t       n_tex;
    const uint8_t *tex_rgba;   /* all textures back to back                                 */
    const uint64_t*tex_offset; /* [n_tex]  byte offset of each texture in tex_rgba          */
    const uint32_t*tex_w;      /* [n_tex]                                                   */
    const uint32_t*tex_h;      /* [n_tex]                                                   */
    /* lamps: scene.h:134-140 */
    uint32_t       n_lamp;
    const float   *lamp_pos;        /* [n_lamp][3]                                          */
    const float   *lamp_intensity;  /* [n_lamp]                                             */
    const float   *lamp_distance;   /* [n_lamp]                                             */
    /* camera and world: scene.h:148-152 */
    float cam_pos[3];
    float cam_mat[9];          /* cam_mat[3*i+c] = JSON transform[i][c]; world[i] = sum_c cam_mat[3*i+c]*cam[c] */
    float horizon[3];          /* worldHorizonColor                                         */
    float ambient[3];          /* worldAmbientColor                                         */
    /* The two Material fields the reference declares but never loads or reads (scene.h:28-29); both may be NULL
     * (= all zero).  Only HIPEXEC_FLAG_REFRACT looks at them. */
    const float   *mat_kt;     /* [n_mat]  refractionFactor: weight of the refracted ray's colour            */
    const float   *mat_ior;    /* [n_mat]  refractionHardness, taken as the relative index of refraction
                                           (values <= 0 count as 1)                                          */
    float ambient_factor;      /* worldAmbientFactor (scene.h:152, scene.cpp:30); only HIPEXEC_FLAG_AO reads it, as the
                                  reference's ENABLE_AO build does (ray_tracer.cpp:423)                     */
} hipexec_scene_desc;

/* Run-time counterpart of the compile-time knobs of config.h:8-20. */
typedef struct hipexec_render_cfg {
    int32_t width;        /* full frame width  (WIDTH)                                      */
    int32_t height;       /* full frame height (HEIGHT)                                     */
    int32_t mode;         /* HIPEXEC_MODE_CPU or HIPEXEC_MODE_CL                            */
    int32_t max_depth;    /* MAX_REFLECTION_DEPTH with ENABLE_REFLECTION; 0 = reflection off.
                             Ignored (treated as 0) in HIPEXEC_MODE_CL.                     */
    int32_t tone;         /* 1: store powf(c/2.2f, 0.3f) per channel (ray_tracer.cpp:47-51);
                             0: store the linear colour                                     */
    int32_t tile_first;   /* first 8-row tile rendered by this call                          */
    int32_t tile_step;    /* stride between rendered tiles (1 = whole frame; G = one of G
                             interleaved shards)                                            */
    int32_t flags;        /* HIPEXEC_FLAG_* bits                                             */
} hipexec_render_cfg;

#define HIPEXEC_FLAG_TIME_KERNELS 1   /* time every kernel with HIP events (fills *_ms in stats) */
/* With HIPEXEC_FLAG_TIME_KERNELS: time only the kernel classes k whose bit HIPEXEC_FLAG_TIME_CLASS(k) is set (none
 * set: all classes).  Every timed launch costs two event records on its stream; a measurement of one class leaves the
 * others undisturbed. */
#define HIPEXEC_FLAG_TIME_CLASS_SHIFT 8
#define HIPEXEC_FLAG_TIME_CLASS(k) (1 << (HIPEXEC_FLAG_TIME_CLASS_SHIFT + (k)))
#define HIPEXEC_FLAG_SCAN         2   /* this call: index-order scans like the reference's loops
                                         (the only algorithm that reports tests_*_any)           */
#define HIPEXEC_FLAG_BVH          4   /* this call: bounding volume hierarchy (the default unless
                                         the environment says HIPEXEC_ALGO=scan)                 */

#define HIPEXEC_FLAG_BILINEAR      8   /* NOT a reference mode: textured hits blend the four texels around
                                         (u - 0.5, v - 0.5) instead of taking tex_image::get's single texel
                                         (tex_image.h:34-44).  Frame entry points only.                     */

#define HIPEXEC_FLAG_REFRACT      16   /* NOT a reference mode (the reference has no refraction, only the two unused Material
                                         fields of scene.h:28-29): in HIPEXEC_MODE_CPU with max_depth > 0 a hit on a material
                                         with refractionFactor > EPS continues with the REFRACTED ray instead of the mirrored
                                         one, weighted by refractionFactor.  With i = d/|d|, N the stored normal, n the
                                         material's index: entering if dot(i,N) < 0 (eta = 1/n, normal N) else leaving
                                         (eta = n, normal -N); k = 1 - eta^2 (1 - cos^2); k < 0 is total internal reflection
                                         and takes the mirrored direction d - 2N(d.N), otherwise the direction is
                                         eta*i + (eta*cos - sqrt(k))*normal.  Shadow rays are unchanged (transmissive
                                         objects still cast full shadows).  Frame entry points only.                    */

#define HIPEXEC_FLAG_AO           32   /* The reference's optional ambient occlusion (ENABLE_AO, ray_tracer.cpp:167-169,406-435; off in
                                         config.h:11) with its libc rand() replaced by a counter-based generator, so the result
                                         is a function of the scene and not of thread schedules: after the ambient colour,
                                         ret += hit.color * ao with ao = (sum over HIPEXEC_AO_RAYS rays i of ambient_factor if the
                                         ray (P, r_i) is unoccluded) / HIPEXEC_AO_RAYS; r_i = (|2 u_{3i}| - 1, |2 u_{3i+1}| - 1,
                                         |2 u_{3i+2}| - 1), negated if dot(r_i, N') < EPS, N' = N or -N facing the viewer
                                         (ray_tracer.cpp:414,426-434); u_c = (h >> 8) * 2^-24 in [0,1) with
                                         h = pcg(pcg(pcg(pcg(c) ^ bits(P.x)) ^ bits(P.y)) ^ bits(P.z)),
                                         pcg(v): v = v*747796405 + 2891336453; w = ((v >> ((v >> 28) + 4)) ^ v) * 277803737;
                                         return (w >> 22) ^ w.  HIPEXEC_MODE_CPU, hierarchy algorithm, at most 32 lamps; its
                                         rays are counted as shadow rays.  Not reference parity (rand() cannot be reproduced). */
#define HIPEXEC_AO_RAYS 30            /* AO_RAYS_COUNT, config.h:14 */

/* Both algorithms evaluate the same exact ray/primitive arithmetic and return identical hits and
 * colours; they differ in which tests they can prove unnecessary. */
enum { HIPEXEC_ALGO_SCAN = 0, HIPEXEC_ALGO_BVH = 1 };

#define HIPEXEC_ABI_VERSION 7

/* Kernel classes reported in hipexec_stats. */
enum {
    HIPEXEC_K_PRIMARY = 0,   /* closest hit of camera rays (hoisted per-triangle constants)  */
    HIPEXEC_K_CLOSEST = 1,   /* closest hit of bounce / user rays                            */
    HIPEXEC_K_ANYHIT  = 2,   /* shadow rays                                                  */
    HIPEXEC_K_SHADE   = 3,   /* shadow-ray generation, shading, resolve, packing             */
    HIPEXEC_K_COUNT   = 4
};

/*
 * Work counters of one call.  A "ray" is one closest-hit or any-hit query.  tests_* is the
 * algorithmic number of ray-primitive tests of the reference's linear scans: n_tri + n_sph
 * per closest-hit ray; (index of the first occluder in scan order) + 1, or n_tri + n_sph
 * when unoccluded, per shadow ray.  tests_tri_any / tests_sph_any are only counted by
 * HIPEXEC_ALGO_SCAN (the hierarchy does not visit triangles in index order) and are 0 otherwise.
 */
typedef struct hipexec_stats {
    uint64_t rays_primary;
    uint64_t rays_bounce;
    uint64_t rays_shadow;
    uint64_t tests_tri_closest;   /* ray-triangle tests of primary + bounce rays            */
    uint64_t tests_sph_closest;
    uint64_t tests_tri_any;       /* ray-triangle tests of shadow rays (reference order)    */
    uint64_t tests_sph_any;
    uint64_t tests_tri_primary;   /* part of tests_tri_closest due to primary rays          */
    uint64_t exact_groups[3];     /* per kernel class (primary, closest, any-hit): number of
                                     (wave, triangle group) pairs that left the filter-only path
                                     for exact tests; a diagnostic of filter selectivity        */
    uint64_t scanned_chunks[3];   /* per kernel class: (wave, 128-triangle chunk) pairs actually scanned;
                                     the rest were skipped by the bounding-box test                */
    uint64_t bvh_nodes[3];        /* per ray class (camera, other closest-hit, shadow): hierarchy
                                     nodes visited, summed over rays (HIPEXEC_ALGO_BVH)            */
    uint64_t bvh_tests[3];        /* per ray class: exact ray-triangle tests evaluated, summed over
                                     rays (HIPEXEC_ALGO_BVH)                                       */
    uint64_t bvh_stack_overflows; /* rays whose traversal stack overflowed and that were finished by a
                                     linear scan instead (same result, slow); expected 0          */
    uint32_t algorithm;           /* HIPEXEC_ALGO_* the call ran with                             */
    uint32_t lanes;               /* concurrent lanes (streams with their own buffers) the frame was split
                                     over: 1, or 2 by default for frames with bounce levels; kernel_ms then
                                     sums overlapping launches and can exceed total_ms                    */
    uint32_t launches[HIPEXEC_K_COUNT];
    float    kernel_ms[HIPEXEC_K_COUNT];   /* summed HIP-event time, only with HIPEXEC_FLAG_TIME_KERNELS */
    float    total_ms;                     /* HIP-event time of the whole call on its stream */
    uint32_t retraced;                     /* times the call was traced again because a launch, sized on the host from
                                              an earlier frame's ray counts, proved smaller than its queue (the results
                                              and the other fields are the last tracing's); 0 for a frame that repeats */
} hipexec_stats;

/* One closest-hit result: the `Hit` of scene.h:48-63 plus the identifiers the reference
 * keeps only implicitly (object index, barycentrics). */
typedef struct hipexec_hit {
    int32_t id;        /* -1 miss; [0,n_tri) triangle index; n_tri + k sphere k             */
    int32_t mat;       /* material index of the hit object or -1                            */
    float   t, u, v;   /* u,v are 0 for spheres and misses                                  */
    float   point[3];
    float   norm[3];
    float   color[3];  /* material colour, or texel-blended colour for textured triangles   */
} hipexec_hit;

/*
 * Replaces OpenClExecutor::OpenClExecutor (opencl_executor.cpp:8-108): selects a device,
 * flattens and uploads the scene once.  `device` is a HIP device ordinal (-1: current).
 */
int hipexec_create(hipexec_t **out, const hipexec_scene_desc *scene, int device);

/* Replaces OpenClExecutor::~OpenClExecutor (opencl_executor.cpp:110-147). NULL is a no-op.
 * The executor's device memory, pinned host memory and streams go back to a pool of the library, from which the next
 * hipexec_create() on that device is served: the reference makes and destroys an executor inside every renderSceneCl
 * call (ray_tracer_cl.cpp:14), and allocation would otherwise cost more than the frame. */
void hipexec_destroy(hipexec_t *h);

/*
 * A further executor of the SAME scene on the same device: it shares the device copy of the scene, the textures and the
 * hierarchy of `src` (no upload, no build; the memory is released when the last executor of the scene is destroyed) and
 * has streams, work buffers, counters and a camera of its own, so its calls may overlap those of `src` and of other
 * clones.  No counterpart in the reference, whose one executor renders one frame at a time (ray_tracer_cl.cpp:14-55);
 * this is how a caller keeps K frames in flight (hipexec_render_device on K executors in turn) without K copies of the
 * scene.  The clone starts with the camera, algorithm and lane count `src` has at that moment.  `src` may be destroyed
 * before its clones.  Set-up calls that may add scene memory (hipexec_set_algorithm to the scans, the first scanning
 * call) must not run on two executors of one scene at the same time.
 */
int hipexec_clone(hipexec_t **out, hipexec_t *src);

/*
 * Camera of later calls of this executor: position and the 3x3 matrix in the layout of hipexec_scene_desc
 * (cam_pos[3], cam_mat[9]); what scene.cpp:121-134 reads from "translate" / "transform".  The reference takes the
 * camera from the Scene it is handed with every call (ray_tracer_cl.cpp:80-92); an executor holds the camera of the
 * scene it was created from until this call changes it.  Nothing of the hierarchy depends on the camera, so the call
 * costs a few stores (with the scan algorithm selected: one pass over the triangles for its camera records).  A frame
 * in flight on this executor is waited for first.
 */
int hipexec_set_camera(hipexec_t *h, const float *cam_pos, const float *cam_mat);

/*
 * Rebuilds the scene's hierarchy with a better topology; for scenes that are kept for many frames.  No counterpart in
 * the reference (it has no hierarchy; its executor lives for one frame, ray_tracer_cl.cpp:14, and hipexec_create's
 * build is made for that: on the device, Morton order, under a millisecond).  This call cuts the triangles top-down by
 * the surface-area heuristic on the host (8 bins per axis, a few threads; 5-20 ms per 100 000 triangles), hands the
 * binary tree to the same device kernels for boxes and 4-wide records, and waits for them.  Rays then visit fewer nodes where the
 * device's Morton order groups badly (5-10 % of a frame on scenes of boxes and walls; nothing measurable on regular terrains,
 * which the device build handles as well); results do not change by a bit, because the hierarchy only decides which
 * exact tests are evaluated and the winner rule is explicit (smallest t, then smallest index).  The arrays that all
 * executors of the scene share (hipexec_clone) are rewritten in place: the call waits for all work on the device first,
 * and no other executor of the scene may be given work until it returns.  A second call, a scene of at most two
 * triangles or one without a hierarchy returns at once.  *build_ms (may be NULL): host + device time of the rebuild.
 */
int hipexec_optimize(hipexec_t *h, float *build_ms);

/*
 * Returns what the library's pool holds for `device` (-1: every device) to the driver.  No counterpart in the
 * reference.  The pool keeps at most 8 GiB per device (environment HIPEXEC_POOL_MB together with HIPEXEC_KNOBS=1
 * changes that, 0 turns the pool off); memory of live executors is not affected.
 */
int hipexec_trim(int device);

/*
 * Use `stream` (a hipStream_t) for all work of later calls instead of the executor's own
 * stream; NULL restores the own stream.  No counterpart in the reference (its queue is
 * private, opencl_executor.cpp:37-41); exists so a caller can order the work with its own
 * HIP work (e.g. an RCCL gather) without a device-wide synchronisation.
 */
int hipexec_set_stream(hipexec_t *h, void *stream);

/*
 * Concurrent lanes a frame with reflection levels is spread over (1..8, default 2; no counterpart in the reference).
 * Lane k of L traces every L-th tile of the call on streams of its own, so that the thinly populated deep levels of
 * one lane run beside the large launches of another.  A single frame at a time is fastest with 2.  For throughput
 * over many frames keep TWO executors of the same scene with ONE lane each and give them the frames in turn
 * (hipexec_render_device is asynchronous; collect a frame with hipexec_sync just before its executor gets the next
 * one): the frames then overlap the same way, by a whole phase, and a process's four hardware queues are not
 * oversubscribed.  Call it before the first frame: lanes that were never used cost nothing.
 */
int hipexec_set_lanes(hipexec_t *h, int lanes);

/*
 * Algorithm (HIPEXEC_ALGO_*) of later calls that do not choose one through
 * hipexec_render_cfg.flags, including the batch entry points.  The reference has one
 * algorithm, the linear scans of ray_tracer.cpp:205-229 / opencl_executor.cpp:110-316
 * (HIPEXEC_ALGO_SCAN keeps their visiting order); HIPEXEC_ALGO_BVH is the default.  A scene
 * for which no hierarchy could be built (non-finite coordinates) is always scanned.
 */
int hipexec_set_algorithm(hipexec_t *h, int algorithm);

/*
 * Replaces renderSceneCl (ray_tracer_cl.cpp:9-56, HIPEXEC_MODE_CL) and renderScene
 * (ray_tracer.cpp:7-54, HIPEXEC_MODE_CPU): renders the tiles selected by cfg into host
 * memory and returns when the pixels are there.
 *   out_rgb : [height][width][3] floats, the image_bitmap layout (image_bitmap.h:22,45),
 *             row 0 first; rows of tiles this call does not render are left untouched.
 *   out_ids : optional [height][width][spp] hit identifiers of the primary samples
 *             (spp = 4 in HIPEXEC_MODE_CPU, 1 in HIPEXEC_MODE_CL), encoding of hipexec_hit.id.
 *   stats   : optional.
 */
int hipexec_render(hipexec_t *h, const hipexec_render_cfg *cfg,
                   float *out_rgb, int32_t *out_ids, hipexec_stats *stats);

/*
 * The frame of hipexec_render delivered in steps of about `rows_per_step` rows (rounded up to
 * whole 8-row tiles, bottom rows of the image first): after each step its rows are in out_rgb
 * and on_rows(user, first_row, row_count) is called (may be NULL).  Counterpart of the
 * progressive display mode, where renderParallel / taskProcessor hand finished 48x48 blocks to
 * the display loop while the rest is still being traced (ray_tracer.cpp:57-153, main.cpp:53-90);
 * SUB_BLOCK_HEIGHT = 48 (config.h:17) corresponds to rows_per_step = 48.  Whole frames only
 * (tile_first 0, tile_step 1).  The pixels are identical to those of hipexec_render.
 */
typedef void (*hipexec_rows_cb)(void *user, int32_t first_row, int32_t row_count);
int hipexec_render_progressive(hipexec_t *h, const hipexec_render_cfg *cfg, int32_t rows_per_step,
                               float *out_rgb, hipexec_rows_cb on_rows, void *user, hipexec_stats *stats);

/*
 * Replaces renderSubBlockCl (ray_tracer_cl.cpp:59-105): renders the pixels [x, x+w) x [y, y+h) of the frame cfg
 * describes (cfg->width / height are the reference's fullWidth / fullHeight, :66-72) into out_rgb, the WHOLE frame's
 * [height][width][3] image; pixels outside the rectangle are left untouched, and the part of the rectangle outside
 * the frame is dropped like image_bitmap::setPixel drops it (image_bitmap.h:36-49).  Camera rays are made on the
 * device from the pixel coordinates (the expressions of :80-92), traced, toned (:94-104) and only the rectangle
 * crosses PCIe.  The pixels equal those hipexec_render gives for the same cfg.  out_ids (optional) is the whole
 * frame's [height][width][spp] array.  Whole frames only (tile_first 0, tile_step 1).
 */
int hipexec_render_rect(hipexec_t *h, const hipexec_render_cfg *cfg, int32_t x, int32_t y, int32_t w, int32_t hgt,
                        float *out_rgb, int32_t *out_ids, hipexec_stats *stats);

/*
 * Same work as hipexec_render but asynchronous and device-resident: results stay in device
 * memory of the executor's device, for a caller that gathers shards itself.
 *   d_out_rgba : device pointer, [n_tiles*8][width][4] floats (RGB + 1.0), the rendered
 *                tiles back to back in increasing tile order.
 *   d_out_ids  : optional device pointer, [n_tiles*8][width][spp] int32.
 * n_tiles = number of k >= 0 with tile_first + k*tile_step < ceil(height/8).
 * The call enqueues on the executor's stream and returns; stats (optional, host memory) are
 * complete only after hipexec_sync().
 */
int hipexec_render_device(hipexec_t *h, const hipexec_render_cfg *cfg,
                          float *d_out_rgba, int32_t *d_out_ids, hipexec_stats *stats);

/* Blocks until all work enqueued by this executor has finished and pending stats are final. */
int hipexec_sync(hipexec_t *h);

/*
 * Replaces computeClosestHitsCl (ray_tracer_cl.cpp:181-242) and with it
 * OpenClExecutor::computeClosestHitTriangle / computeClosestHitSphere
 * (opencl_executor.cpp:149-264, 341-449): closest hit, triangle/sphere merge, material and
 * texture lookup for a batch of rays.  rays6 is host memory, n rays of {p_x,p_y,p_z,d_x,d_y,d_z}
 * (the RayData packing of scene.h:156-169).  n = 0 is legal.
 */
int hipexec_closest_hits(hipexec_t *h, int mode, const float *rays6, uint32_t n,
                         hipexec_hit *out);

/*
 * Replaces computeAnyHitsCl (ray_tracer_cl.cpp:245-262) and OpenClExecutor::computeAnyHitTriangle /
 * computeAnyHitSphere (opencl_executor.cpp:266-339, 451-524).  hits[i] is OR-ed with the
 * occlusion result like the reference does (caller pre-initialises).
 */
int hipexec_any_hits(hipexec_t *h, int mode, const float *rays6, uint32_t n, char *hits);

/*
 * hipexec_any_hits for callers that cannot allocate.  No counterpart in the reference; it exists for
 * callers that may not reach hipMalloc / hipFree: code under HIP graph capture, tight serving loops,
 * harnesses that forbid allocation.  Same occlusion test as hipexec_any_hits, but
 *   - d_rays6 and d_hits are DEVICE pointers on the executor's device (n rays of 6 floats; n chars,
 *     OR-ed into like hipexec_any_hits does);
 *   - the work is enqueued on the executor's stream and the call returns without waiting
 *     (hipexec_sync() or the caller's own stream ordering makes the result visible);
 *   - the ray records, occlusion words and work counters the kernels need live in a fixed device
 *     array of the library (about 9 MiB per GPU, held until the process ends) instead of grown
 *     buffers, so the call performs no hipMalloc, hipFree or other allocating call;
 *   - n must not exceed HIPEXEC_STATIC_MAX_RAYS (HIPEXEC_E_LIMIT otherwise).
 * All calls on one GPU share that array: calls that overlap on one GPU (two executors, or two
 * streams) race; one executor's calls are ordered by its stream.
 */
#define HIPEXEC_STATIC_MAX_RAYS (1u << 18)
int hipexec_any_hits_static(hipexec_t *h, int mode, const float *d_rays6, uint32_t n, char *d_hits);

/*
 * Replaces traceRaysCl (ray_tracer_cl.cpp:108-178, HIPEXEC_MODE_CL) / traceRay
 * (ray_tracer.cpp:156-201, HIPEXEC_MODE_CPU) for a batch of rays: colours3[i] ([n][3]
 * floats, host) is accumulated into (+=) as the reference does, caller pre-zeros.
 * max_depth as in hipexec_render_cfg.
 */
int hipexec_trace_rays(hipexec_t *h, int mode, int max_depth, const float *rays6, uint32_t n,
                       float *colours3, hipexec_stats *stats);

/* Message of the calling thread's last failed call ("" if none).  Mirrors the reference's
 * diagnostics on std::cerr (opencl_executor.h:67-72) without printing. */
const char *hipexec_last_error(void);

/* Library ABI version; bumped when a struct layout changes. */
int hipexec_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* HIP_EXECUTOR_H */
