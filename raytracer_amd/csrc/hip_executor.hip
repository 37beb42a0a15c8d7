// hip_executor.hip — host side of the C ABI declared in include/hip_executor.h.
//
// Owns the device copy of the scene, the ray/hit/colour work buffers and the per-frame
// sequence of kernel launches.  Build with -ffp-contract=off (see rt_device.hpp).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <climits>
#include <cstring>
#include <new>
#include <string>
#include <array>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <map>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <limits>
#include <vector>

#include "hip_executor.h"
#include "rt_kernels.hpp"
#include "rt_bvh.hpp"
#include "rt_bvh_host.hpp"
#include <hipcub/hipcub.hpp>

using namespace rt;

namespace {

thread_local char g_err[512] = "";

// Experiment and test knobs come from the environment, but only when HIPEXEC_KNOBS=1 is set as well: a production
// process does not change behaviour because of a stray variable.
const char *knob(const char *name)
{
    static const bool enabled = [] { const char *e = getenv("HIPEXEC_KNOBS"); return e && e[0] == '1'; }();
    return enabled ? getenv(name) : nullptr;
}

// HIPEXEC_TRACE_CREATE=1 (with HIPEXEC_KNOBS=1): host time of the stages of hipexec_create / hipexec_destroy on stderr
struct StageTrace {
    bool on; std::chrono::steady_clock::time_point t0, last;
    StageTrace() : on(false) { const char *e = knob("HIPEXEC_TRACE_CREATE"); on = e && e[0] == '1'; t0 = last = std::chrono::steady_clock::now(); }
    void mark(const char *what) {
        if (!on) return;
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[hipexec] %-28s +%7.1f us (at %8.1f)\n", what, std::chrono::duration<double, std::micro>(now - last).count(),
                std::chrono::duration<double, std::micro>(now - t0).count());
        last = now;
    }
};
thread_local StageTrace *g_stage = nullptr;
#define STAGE(what) do { if (g_stage) g_stage->mark(what); } while (0)

int fail(int code, const char *fmt, ...)
{
    va_list ap; va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIPCHK(expr)                                                                             \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail(e_ == hipErrorOutOfMemory ? HIPEXEC_E_NOMEM : HIPEXEC_E_DEVICE,          \
                        "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

#define CHK(expr) do { int r_ = (expr); if (r_ != HIPEXEC_OK) return r_; } while (0)

// grow-only device buffer
// ---------------------------------------------------------------------------------------------------------
// Process-wide pool of device memory, pinned host memory and streams.  The reference builds and destroys its executor
// inside every renderSceneCl call (ray_tracer_cl.cpp:14), and so does the drop-in: with hipMalloc / hipFree /
// hipHostMalloc / hipStreamCreate paid every time, create + destroy cost more than the frame.  Freed blocks are kept per
// device and size class (sizes round up to m * 2^k, m = 8..15: at most 12.5 % over) and handed to the next executor;
// beyond HIPEXEC_POOL_MB (default 8192) of cached bytes per device, or with HIPEXEC_POOL_MB=0, blocks go back to the
// driver.  hipexec_trim() returns everything that is cached.  What the pool holds at process exit is left to the driver.
// Blocks are reused without clearing: every consumer writes what it reads (as with fresh hipMalloc memory).
// ---------------------------------------------------------------------------------------------------------
struct MemPool {
    struct Side { std::multimap<size_t, void *> free; std::unordered_map<void *, size_t> size; size_t cached = 0; };
    struct Dev { Side dev, pin; std::vector<hipStream_t> streams; };
    std::mutex m;
    Dev d[64];
    size_t cap_bytes;
    MemPool() { const char *e = knob("HIPEXEC_POOL_MB"); cap_bytes = (size_t)(e ? std::max(0L, atol(e)) : 8192L) << 20; }
    static size_t size_class(size_t bytes) {
        if (bytes <= 4096) return 4096;
        int k = 63 - __builtin_clzll((unsigned long long)bytes);          // 2^k <= bytes
        size_t step = (size_t)1 << (k - 3);
        return (bytes + step - 1) / step * step;
    }
};
MemPool &pool() { static MemPool *p = new MemPool(); return *p; }           // never destroyed: no ordering against the HIP runtime's teardown

int pool_alloc(int device, bool pinned, size_t bytes, void **out)
{
    MemPool &P = pool();
    const size_t cls = MemPool::size_class(bytes);
    if (device >= 0 && device < 64) {
        std::lock_guard<std::mutex> lk(P.m);
        MemPool::Side &sd = pinned ? P.d[device].pin : P.d[device].dev;
        auto it = sd.free.find(cls);
        if (it != sd.free.end()) { *out = it->second; sd.free.erase(it); sd.cached -= cls; return HIPEXEC_OK; }
    }
    void *p = nullptr;
    hipError_t e = pinned ? hipHostMalloc(&p, cls, hipHostMallocDefault) : hipMalloc(&p, cls);
    if (e != hipSuccess && device >= 0 && device < 64) {                        // out of memory: give the cache back and try once more
        {
            std::lock_guard<std::mutex> lk(P.m);
            MemPool::Side &sd = pinned ? P.d[device].pin : P.d[device].dev;
            for (auto &kv : sd.free) { if (pinned) (void)hipHostFree(kv.second); else (void)hipFree(kv.second); sd.size.erase(kv.second); }
            sd.free.clear(); sd.cached = 0;
        }
        (void)hipGetLastError();
        e = pinned ? hipHostMalloc(&p, cls, hipHostMallocDefault) : hipMalloc(&p, cls);
    }
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? HIPEXEC_E_NOMEM : HIPEXEC_E_DEVICE, "%s of %zu bytes: %s",
                                     pinned ? "hipHostMalloc" : "hipMalloc", cls, hipGetErrorString(e));
    if (device >= 0 && device < 64) {
        std::lock_guard<std::mutex> lk(P.m);
        (pinned ? P.d[device].pin : P.d[device].dev).size[p] = cls;
    }
    *out = p;
    return HIPEXEC_OK;
}

void pool_free(int device, bool pinned, void *p)
{
    if (!p) return;
    MemPool &P = pool();
    if (device >= 0 && device < 64) {
        std::lock_guard<std::mutex> lk(P.m);
        MemPool::Side &sd = pinned ? P.d[device].pin : P.d[device].dev;
        auto it = sd.size.find(p);
        if (it != sd.size.end() && sd.cached + it->second <= P.cap_bytes) { sd.free.emplace(it->second, p); sd.cached += it->second; return; }
        if (it != sd.size.end()) sd.size.erase(it);
    }
    if (pinned) (void)hipHostFree(p); else (void)hipFree(p);
}

int pool_stream(int device, hipStream_t *out)
{
    MemPool &P = pool();
    if (device >= 0 && device < 64) {
        std::lock_guard<std::mutex> lk(P.m);
        std::vector<hipStream_t> &v = P.d[device].streams;
        if (!v.empty()) { *out = v.back(); v.pop_back(); return HIPEXEC_OK; }
    }
    HIPCHK(hipStreamCreateWithFlags(out, hipStreamNonBlocking));
    return HIPEXEC_OK;
}

// the stream must be idle (its owner synchronised it)
void pool_stream_back(int device, hipStream_t s)
{
    if (!s) return;
    MemPool &P = pool();
    if (device >= 0 && device < 64 && P.cap_bytes) {
        std::lock_guard<std::mutex> lk(P.m);
        if (P.d[device].streams.size() < 32) { P.d[device].streams.push_back(s); return; }
    }
    (void)hipStreamDestroy(s);
}

void pool_trim(int device)
{
    MemPool &P = pool();
    std::lock_guard<std::mutex> lk(P.m);
    for (int dv = 0; dv < 64; dv++) {
        if (device >= 0 && dv != device) continue;
        MemPool::Dev &D = P.d[dv];
        if (D.dev.free.empty() && D.pin.free.empty() && D.streams.empty()) continue;
        (void)hipSetDevice(dv);
        for (auto &kv : D.dev.free) { (void)hipFree(kv.second); D.dev.size.erase(kv.second); }
        for (auto &kv : D.pin.free) { (void)hipHostFree(kv.second); D.pin.size.erase(kv.second); }
        D.dev.free.clear(); D.pin.free.clear(); D.dev.cached = D.pin.cached = 0;
        for (hipStream_t st : D.streams) (void)hipStreamDestroy(st);
        D.streams.clear();
    }
}

struct Buf {
    void *p = nullptr; size_t cap = 0; int device = -1;
    int reserve(size_t bytes) {
        if (bytes <= cap) return HIPEXEC_OK;
        if (device < 0) HIPCHK(hipGetDevice(&device));
        if (p) {                                  // growing: work queued on the old block must be over before another owner can get it
            HIPCHK(hipDeviceSynchronize());
            pool_free(device, false, p); p = nullptr; cap = 0;
        }
        size_t want = MemPool::size_class(bytes + 256);
        CHK(pool_alloc(device, false, want, &p));
        cap = want;
        return HIPEXEC_OK;
    }
    void release() { if (p) pool_free(device, false, p); p = nullptr; cap = 0; }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

// Host thread of a further lane.  It lives as long as its executor: starting a std::thread per frame
// delayed the lane's first kernel by 0.2 ms.  After a job the thread polls for the next one for a
// short while (a render loop submits the next frame within that time) before it blocks.
struct LaneWorker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::function<void()> job;
    std::atomic<int> state{0};               // 0 idle, 1 job posted, 2 running
    bool quit = false;
    void run() {
        for (;;) {
            for (int spin = 0; spin < 20000 && state.load(std::memory_order_acquire) != 1; spin++) {
#if defined(__x86_64__)
                __builtin_ia32_pause();
#endif
            }
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return state.load(std::memory_order_acquire) == 1 || quit; });
                if (quit) return;
                state.store(2, std::memory_order_release);
            }
            job();
            {
                std::lock_guard<std::mutex> lk(m);
                state.store(0, std::memory_order_release);
            }
            cv.notify_all();
        }
    }
    void submit(std::function<void()> j) {
        if (!th.joinable()) th = std::thread([this] { run(); });
        {
            std::lock_guard<std::mutex> lk(m);
            job = std::move(j);
            state.store(1, std::memory_order_release);
        }
        cv.notify_all();
    }
    void wait() {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return state.load(std::memory_order_acquire) == 0; });
    }
    void stop() {
        if (!th.joinable()) return;
        wait();
        { std::lock_guard<std::mutex> lk(m); quit = true; }
        cv.notify_all();
        th.join();
    }
};

struct TimedLaunch { hipEvent_t a, b; int klass; };

constexpr uint64_t SHADOW_QUEUE_MAX_ENTRIES = 96ull << 20;   // 3 GiB of 32-byte shadow rays per batch
constexpr uint32_t BATCH_API_CHUNK = 4u << 20;               // rays per chunk of the batch entry points
constexpr uint64_t BATCH_MAX_SAMPLES = 64ull << 20;          // samples per batch of a frame without a shadow queue
constexpr uint32_t TAIL_MAX_RAYS = 300000u;                    // see use_tail
constexpr int N_CNT64 = 17;                                  // device counters in use per stripe, see call_end
constexpr size_t CNT64_WORDS = (size_t)CNT_STRIPES * CNT_STRIDE;   // all stripes
static_assert(N_CNT64 <= (int)CNT_STRIDE, "counters fit a stripe");
constexpr int LV_SLOTS = 8;                                  // batches of one call (per lane) whose level counts are kept apart
constexpr int LV_PER = 68;                                   // level counters per batch: entry L = rays queued for recursion level L (max_depth <= 64)

// Ray-queue sizes of the recursion levels of one batch, as the last frame with the same key produced them, kept per
// executor lane.  No frame of the hierarchy algorithm waits for a count to come back from the device: every launch of
// a level is sized on the host, the kernels clamp to the queue's true fill on the device (queue_fill), and a launch
// that is larger than its queue costs only blocks that exit at once.  The sizes come from
//   * the bound n(L+1) <= n(L) (a ray has at most one successor) when nothing is remembered: first frame of a
//     configuration, or the repeat of a frame whose sizes proved too small;
//   * otherwise the sizes the last frame with this key produced plus a margin (a quarter, twice the change from the
//     frame before that, 4096 rays), capped by the bound: a frame that repeats and a camera that moves are sized alike.
// The sizes the kernels produced come back with the frame and are compared with what was launched when the call is
// collected: only a level that was launched too SMALL has lost rays, and only then the frame is traced again (from
// the bound, which cannot be too small).  Decisions that depend on a launch size (threads per block, tail kernel,
// side stream) change speed, never bits.
struct LevelPred {
    std::array<int64_t, 24> key{};
    std::vector<uint32_t> n;                                  // n[L], L >= 1: queue sizes the last frame with this key produced
    // Levels 1..seen were counted by that frame itself (entries past the end of n are true zeros).  A frame that hands
    // its small deep levels to k_bvh_tail counts only up to the level the tail kernel starts at (the kernel keeps the
    // later rays to itself); entries above `seen` are then carried over from the last frame that did count them, and
    // a level of which nothing is known is launched for the bound -- never for "zero rays", which is what made a
    // frame sized with the margin (too large for the tail kernel at that level) lose the next level's rays and be
    // traced twice.
    int seen = 0;
    std::vector<uint32_t> delta;                              // delta[L]: by how much n[L] differed from the frame before (levels both counted)
    bool steady = false;                                      // the frame before produced the same sizes on the levels both counted
};
struct LevelCheck { int slot; std::array<int64_t, 24> key; std::vector<uint32_t> launched; int tail_level; };
// How the launches of a batch's levels get their sizes.
enum { SIZE_AUTO = 0,      // from remembered sizes where they exist, else from the bound
       SIZE_BOUND = 1,     // from the bound n(L+1) <= n(L) only (repeat of a frame; HIPEXEC_PREDICT=2)
       SIZE_WAIT = 2 };    // every size is read back from the device before the next launch (scan algorithm, batch calls, HIPEXEC_PREDICT=0)

// Fixed work memory of hipexec_any_hits_static (one copy per GPU, held until the process ends):
// [0, STATIC_CNT_BYTES) the striped work counters (CNT_STRIPES x CNT_STRIDE 64-bit words = 16 KiB), then the ray
// records (32 bytes per ray), then one occlusion word per ray.
constexpr size_t STATIC_CNT_BYTES = CNT64_WORDS * sizeof(unsigned long long);
constexpr size_t STATIC_REC_BYTES = (size_t)HIPEXEC_STATIC_MAX_RAYS * 32;
constexpr size_t STATIC_OCC_BYTES = (size_t)HIPEXEC_STATIC_MAX_RAYS * 4;
constexpr size_t STATIC_BYTES = STATIC_CNT_BYTES + STATIC_REC_BYTES + STATIC_OCC_BYTES;
constexpr int STATIC_MAX_DEVICES = 64;

}  // namespace

// In host code this name is only a handle for hipGetSymbolAddress: the address the kernels use is
// g_static_ptr[device], looked up once per device by static_scratch_init().
static __device__ __align__(256) unsigned char g_static_scratch[STATIC_BYTES];

namespace {
unsigned char *g_static_ptr[STATIC_MAX_DEVICES] = {};
std::mutex g_static_mutex;
}  // namespace

// Device copy of a scene and its hierarchy: owned jointly by the executor that uploaded it and the executors cloned from
// it (hipexec_clone), released by the last of them.  Everything in it is read-only once built, except that the first
// executor that scans adds the camera-independent scan records (under the mutex).
struct SceneShare {
    std::atomic<int> refs{1};
    int device = 0;
    std::mutex m;                             // arena and shared scan records
    std::vector<void *> allocs;               // chunks of the scene arena
    char *arena = nullptr; size_t arena_cap = 0, arena_used = 0;
    float *d_tri_p = nullptr, *d_tri_e1 = nullptr, *d_tri_e2 = nullptr;   // the caller's triangle arrays on the device (set-up kernels read them)
    std::vector<float> lamp_pos_host;         // lamp positions, for the lamp records of the scans
    bool scan_shared = false;                 // pair_gen / pair_lamp / chunk_box exist
    bool optimized = false;                   // the hierarchy has the surface-area topology (hipexec_optimize)
    uint32_t popular[sah_detail::CELL_WORDS] = {};   // normal cells that hold a fifth of the triangle area (orientation classes)
    const PairGen *pair_gen = nullptr; const PairCam *pair_lamp = nullptr; const float4 *chunk_box = nullptr;
    // remembered queue sizes (newest last, at most 64), common to the executors of the scene and their lanes: with
    // several frames in flight each executor sizes its next frame from the newest frame any of them has collected
    std::mutex pred_m;
    std::vector<LevelPred> preds;
};

struct hipexec {
    int device = 0;
    SceneShare *share = nullptr;              // null for helper lanes (they live inside their executor)
    SceneShare *pred_home = nullptr;          // whose remembered queue sizes this lane uses (its executor's scene)
    hipStream_t own_stream = nullptr, stream = nullptr;
    DScene S{};
    DBvh B{};
    bool bvh_ok = false;                      // a hierarchy exists for this scene
    bool bvh_default = true;                  // algorithm of calls that do not choose (HIPEXEC_ALGO=scan|bvh)
    bool use_bvh = false;                     // algorithm of the call in progress
    int closest_per_block = 0;                // rays owned by a block of k_bvh_closest_p; 0 = by ray class (HIPEXEC_CLOSEST_PER_BLOCK)
    int sched_w = 8;                          // step scheduling weight of bvh_burst (HIPEXEC_SCHED_W), in eighths
    int refill_closest = REFILL_BELOW;        // HIPEXEC_REFILL_CLOSEST: live lanes below which a wave fetches new rays
    bool bvh_refill = true;                   // closest-hit kernel that refills idle lanes (HIPEXEC_BVH_REFILL=0: one ray per lane)
    uint32_t lamp_words = 1;
    hipStream_t upload_stream = nullptr;      // stream of upload() while it is not the executor's (scene set-up: beside the hierarchy build)
    void *build_tmp = nullptr;                // temporaries of the hierarchy build, until the build has been waited for
    int morton_minor = -1;                    // axis whose code bits come last in the device build (-1: none), see build_scene
    bool scan_ready = false;                  // the scan algorithm's records exist and belong to this executor's camera (ensure_scan_records)
    Buf cam_primx, cam_pairs;                 // the camera-dependent scan records of this executor (k_setup_cam_pairs)
    Buf rayA, rayB, hit0, hitN, hitM, local, occl, shadowq, out_rgba, out_ids, stage_in, stage_out;
    uint32_t *d_cnt32 = nullptr;              // [0] next-level queue fill, [1 + l] shadow queue fill of lamp l
    uint32_t n_cnt32 = 0;
    bool closest_lds = false, anyhit_lds = false;
    int closest_r = 3, anyhit_r = 2;          // rays per lane (tunable through HIPEXEC_CLOSEST_R / HIPEXEC_ANYHIT_R)
    unsigned long long *d_cnt64 = nullptr;    // [0] any-hit triangle tests, [1] any-hit sphere tests
    uint32_t *h_cnt32 = nullptr;              // pinned
    uint32_t *d_lvl = nullptr, *h_lvl = nullptr;   // [LV_SLOTS][LV_PER] level queue fills of the call in progress (device / pinned)
    int lv_slot = 0;                          // next free slot of the call in progress
    std::vector<LevelCheck> checks;           // batches of the call in progress that ran from remembered sizes
    int size_default = SIZE_AUTO;             // HIPEXEC_PREDICT=0: SIZE_WAIT, =2: SIZE_BOUND
    int size_mode = SIZE_AUTO;                // of the call in progress (SIZE_BOUND while a frame is being repeated, SIZE_WAIT for batch calls)
    int predict_spoil = 0;                    // test hook: remembered level-1 sizes are stored this much too large (or too small)
    bool redo = false;                        // a level was launched too small: the frame has to be repeated
    uint32_t retraced = 0u;                   // 1 while the repeat of such a frame runs (reported in the call's statistics)
    hipexec_render_cfg pending_cfg{};         // arguments of an asynchronous call, for such a repeat
    float4 *pending_rgba = nullptr; int32_t *pending_ids = nullptr;
    unsigned long long *h_cnt64 = nullptr;    // pinned
    std::vector<TimedLaunch> timed;
    std::vector<hipEvent_t> event_pool;
    uint32_t timing = 0;                      // bit k: launches of kernel class k are bracketed by HIP events
    bool timed_this = false;                  // the launch between timed_begin and timed_end is being timed
    hipexec_stats *pending_stats = nullptr;
    hipexec_stats acc{};
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    bool call_open = false;
    // Second lane: an executor of its own (stream, work buffers, counters) on the same scene.  A frame
    // is split into even and odd tiles that the two lanes trace concurrently, so that the thinly
    // populated tail of one lane's kernels (a few very long rays) overlaps the other lane's work.
    int win_lo = 0, win_hi = INT_MAX;         // window into the call's tile list (progressive rendering, sub-rectangles)
    int rect[4] = {0, 0, INT_MAX, INT_MAX};   // pixel window x0, y0, x1, y1 of the call (hipexec_render_rect), clipped to the frame when used
    std::vector<hipexec *> helpers;           // further lanes of this executor (own streams, buffers, counters; no scene memory)
    std::vector<hipexec *> prog_clones;       // executors of the same scene that hipexec_render_progressive keeps steps in flight on
    void *prog_pin = nullptr; size_t prog_pin_cap = 0;   // pinned staging of this executor's step of a progressive frame
    bool is_helper = false;
    LaneWorker *worker = nullptr;             // host thread of a helper lane
    int lanes_used = 1;                       // lanes of the call in progress
    int max_lanes = 2;                        // HIPEXEC_LANES
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // side stream of this lane: lights a level (shadow rays, shading) while the next level is traced on `stream`
    hipStream_t side = nullptr;
    hipEvent_t ev_hits[2] = {nullptr, nullptr}, ev_lit[2] = {nullptr, nullptr};
    int fuse_shade = 1;                       // HIPEXEC_FUSE_SHADE: 0 k_shade always a launch of its own, 1 by batch, 2 always fused
    uint32_t uniform_fetch = 3u;              // HIPEXEC_UNIFORM_FETCH: bit 0 shadow rays, bit 1 bounce and batch closest-hit rays; a clear bit = every inner-node step of that class fetches per lane
    bool packet = true;                       // HIPEXEC_PACKET=0: per-lane traversal for camera rays too
    bool fuse_raygen = true;                  // HIPEXEC_FUSE_RAYGEN=0: camera rays from k_raygen, a launch of its own
    uint32_t shadow_waves6_from = 1u;         // HIPEXEC_SHADOW_WAVES6_FROM: lamps from which the shadow kernel compiled for RT_SHADOW_WAVES_MANY waves per SIMD is used (one lamp, C4: -2 % against the five-wave variant)
    uint32_t simple_from = 0u;                // HIPEXEC_SIMPLE_FROM: closest-hit launches of at least this many rays take one ray per lane
                                              // (k_bvh_closest); smaller ones the kernel whose lanes fetch a new ray when theirs is decided
    bool shadow_direct = true;                // HIPEXEC_SHADOW_DIRECT=0: shadow rays through the block's list in LDS (k_bvh_shadow)
    uint32_t shadow_per_block = 256u;         // HIPEXEC_SHADOW_PER_BLOCK: samples owned by a block of k_bvh_shadow (multiple of 64, <= 256)
    uint32_t tail_max = TAIL_MAX_RAYS;        // HIPEXEC_TAIL_MAX: largest ray queue handed to k_bvh_tail (0: never)
    int block_closest = 256, block_shadow = 256, block_tail = 256;   // HIPEXEC_BLOCK_CLOSEST / _SHADOW / _TAIL: threads per block of the traversal kernels
    uint32_t block64_below = 2500000u;        // HIPEXEC_BLOCK64_BELOW: launches of at most this many rays take one wave per block
    int overlap_levels = 1;                   // HIPEXEC_OVERLAP: 0 never, 1 for batches up to OVERLAP_MAX_SAMPLES, 2 always
};

namespace {

// Address of this device's copy of g_static_scratch; `device` must be the current device.
int static_scratch_init(int device)
{
    if (device < 0 || device >= STATIC_MAX_DEVICES) return HIPEXEC_OK;       // such a device has no static entry point
    std::lock_guard<std::mutex> lk(g_static_mutex);
    if (g_static_ptr[device]) return HIPEXEC_OK;
    void *p = nullptr;
    HIPCHK(hipGetSymbolAddress(&p, HIP_SYMBOL(g_static_scratch)));
    if (!p) return fail(HIPEXEC_E_DEVICE, "hipGetSymbolAddress returned a null address for device %d", device);
    g_static_ptr[device] = static_cast<unsigned char *>(p);
    return HIPEXEC_OK;
}

// Scene memory comes from a few large chunks (one hipMalloc / hipFree per chunk instead of one per
// array: creating and destroying an executor per frame, as ray_tracer_cl.cpp:14 does, stays cheap).
int scene_alloc(hipexec *h, size_t bytes, void **out)
{
    constexpr size_t CHUNK = 32u << 20, ALIGN = 256;
    bytes = (bytes + ALIGN - 1) / ALIGN * ALIGN;
    SceneShare *sh = h->share;
    if (!sh) return fail(HIPEXEC_E_INVALID, "a helper lane owns no scene memory");
    std::lock_guard<std::mutex> lk(sh->m);
    if (sh->arena_used + bytes > sh->arena_cap) {
        size_t want = std::max(bytes, CHUNK);
        void *p = nullptr;
        CHK(pool_alloc(h->device, false, want, &p));
        sh->allocs.push_back(p);
        sh->arena = static_cast<char *>(p); sh->arena_cap = want; sh->arena_used = 0;
    }
    *out = sh->arena + sh->arena_used;
    sh->arena_used += bytes;
    return HIPEXEC_OK;
}

// drops one reference; the last one hands the scene's memory back (the caller has synchronised its streams)
void share_release(SceneShare *sh)
{
    if (!sh) return;
    if (sh->refs.fetch_sub(1, std::memory_order_acq_rel) != 1) return;
    for (void *p : sh->allocs) pool_free(sh->device, false, p);
    delete sh;
}

// Device array of max(count, min_count) elements: the first `count` copied from src (if any), the
// rest zero.  `src` must stay valid until the stream is synchronised (build_scene does that before
// it returns).  Everything goes through the executor's stream: its set-up kernels run there and a
// non-blocking stream is not ordered against the null stream.
template <class T>
int upload(hipexec *h, const T *src, size_t count, T **dst, size_t min_count = 1)
{
    size_t n = std::max(count, min_count);
    void *p = nullptr;
    CHK(scene_alloc(h, n * sizeof(T), &p));
    size_t copied = (count && src) ? count : 0;
    hipStream_t st = h->upload_stream ? h->upload_stream : h->stream;
    if (copied) HIPCHK(hipMemcpyAsync(p, src, copied * sizeof(T), hipMemcpyHostToDevice, st));
    if (copied < n) HIPCHK(hipMemsetAsync(static_cast<char *>(p) + copied * sizeof(T), 0, (n - copied) * sizeof(T), st));
    *dst = reinterpret_cast<T *>(p);
    return HIPEXEC_OK;
}

int validate_scene(const hipexec_scene_desc *s)
{
    if (!s) return fail(HIPEXEC_E_INVALID, "scene descriptor is NULL");
    if (s->n_tri && (!s->tri_p || !s->tri_e1 || !s->tri_e2 || !s->tri_norm))
        return fail(HIPEXEC_E_INVALID, "triangle arrays missing");
    if (s->n_sph && (!s->sph_center || !s->sph_radius)) return fail(HIPEXEC_E_INVALID, "sphere arrays missing");
    if (s->n_mat && (!s->mat_color || !s->mat_kd || !s->mat_ks || !s->mat_hard || !s->mat_kr))
        return fail(HIPEXEC_E_INVALID, "material arrays missing");
    if (s->n_lamp && (!s->lamp_pos || !s->lamp_intensity || !s->lamp_distance))
        return fail(HIPEXEC_E_INVALID, "lamp arrays missing");
    if (s->n_tex && (!s->tex_rgba || !s->tex_offset || !s->tex_w || !s->tex_h))
        return fail(HIPEXEC_E_INVALID, "texture arrays missing");
    if (s->n_tri > 0x7ff00000u || s->n_sph > 0x00100000u) return fail(HIPEXEC_E_LIMIT, "too many primitives");
    for (uint32_t i = 0; i < s->n_tri && s->tri_mat; i++)
        if (s->tri_mat[i] < -1 || s->tri_mat[i] >= (int32_t)s->n_mat)
            return fail(HIPEXEC_E_INVALID, "triangle %u: material index %d out of range", i, s->tri_mat[i]);
    for (uint32_t i = 0; i < s->n_sph && s->sph_mat; i++)
        if (s->sph_mat[i] < -1 || s->sph_mat[i] >= (int32_t)s->n_mat)
            return fail(HIPEXEC_E_INVALID, "sphere %u: material index %d out of range", i, s->sph_mat[i]);
    for (uint32_t i = 0; i < s->n_mat && s->mat_tex; i++)
        if (s->mat_tex[i] < -1 || s->mat_tex[i] >= (int32_t)s->n_tex)
            return fail(HIPEXEC_E_INVALID, "material %u: texture index %d out of range", i, s->mat_tex[i]);
    return HIPEXEC_OK;
}

// Hierarchy over the triangles (rt_bvh.hpp).  Scenes with non-finite or astronomically large
// coordinates, or too many triangles for the leaf encoding, get none and are always scanned.
// `sah` false: the Morton-order radix tree, all on the device (lo / hi: the scene's bounds).  `sah` true: the
// topology comes from the surface-area build on the host (rt_bvh_host.hpp; waits for the stream, reads the triangle
// records back) and the arrays of an earlier build are written again in place: nothing may be traversing them.
int build_bvh(hipexec *h, const float4 *tri_exact, const double *lo, const double *hi, bool all_finite, bool sah = false)
{
    const DScene &S = h->S;
    DBvh &B = h->B;
    const uint32_t n = S.n_tri;
    float4 *tris = nullptr, *nodes = nullptr; uint32_t *ids = nullptr;
    if (!sah) {
        B.n = n; B.root = BVH_NONE; B.nodes = nullptr; B.tris = nullptr; B.ids = nullptr;
        B.stack_limit = BVH_STACK;
        if (const char *e = knob("HIPEXEC_BVH_STACK_LIMIT")) B.stack_limit = std::max(0, std::min(BVH_STACK, atoi(e)));   // test hook
        B.scene_mag = std::fabs(S.center[0]) + std::fabs(S.center[1]) + std::fabs(S.center[2]) + S.rmax;
        // clamp of the slab test's reciprocals (tray_make): at least 1e12 * size^2, from the shortest direction that can pass
        // the exact tests' |det| > 1e-4 with edges of at most 2 sqrt(3) times the scene's size; none where that leaves the float range
        {
            const double want = std::max(1.0e30, 1.0e12 * (double)B.scene_mag * (double)B.scene_mag);
            B.inv_clamp = want < 3.0e38 ? (float)want : std::numeric_limits<float>::infinity();
        }
        h->bvh_ok = all_finite && n <= BVH_MAX_TRIS;
        CHK(upload<float4>(h, nullptr, 0, &tris, 3 * (size_t)n + 3));      // + one record: a one-triangle leaf is fetched as two
        CHK(upload<uint32_t>(h, nullptr, 0, &ids, (size_t)n + 1));
        CHK(upload<float4>(h, nullptr, 0, &nodes, BVH_NODE_F4 * (size_t)n + 1));
        B.tris = tris; B.ids = ids; B.nodes = nodes;
        if (!h->bvh_ok || n == 0) return HIPEXEC_OK;
        B.root = (n <= (uint32_t)BVH_LEAF_MAX) ? (BVH_LEAF | (n << 28)) : 0u;
    } else {
        tris = const_cast<float4 *>(B.tris); ids = const_cast<uint32_t *>(B.ids); nodes = const_cast<float4 *>(B.nodes);
    }

    // temporaries: one block, carved up, released at the end
    unsigned long long *keys_a = nullptr, *keys_b = nullptr;
    uint32_t *child = nullptr, *parent = nullptr, *range = nullptr, *arrived = nullptr, *top_list = nullptr;
    float4 *box_lo = nullptr, *box_hi = nullptr, *nrm_lo = nullptr, *nrm_hi = nullptr;
    void *sort_tmp = nullptr;
    size_t tmp_bytes = 0;
    HIPCHK(hipcub::DeviceRadixSort::SortKeys(nullptr, tmp_bytes, keys_a, keys_b, (int)n, 0, 63, h->stream));
    char *block = nullptr;
    size_t used = 0;
    auto carve = [&](size_t bytes) -> void * { void *p = block + used; used += (bytes + 255) / 256 * 256; return p; };
    const size_t total = 2 * (8 * (size_t)n + 256) + 3 * (8 * (size_t)n + 256) + 2 * (4 * (size_t)n + 512) + 4 * (32 * (size_t)n + 256) + tmp_bytes + 256;
    CHK(pool_alloc(h->device, false, total, (void **)&block));
    int rc = [&]() -> int {
        keys_a = (unsigned long long *)carve(8 * (size_t)n); keys_b = (unsigned long long *)carve(8 * (size_t)n);
        child = (uint32_t *)carve(8 * (size_t)n); parent = (uint32_t *)carve(8 * (size_t)n); range = (uint32_t *)carve(8 * (size_t)n);
        arrived = (uint32_t *)carve(4 * (size_t)n + 4);          // + the count of top_list
        top_list = (uint32_t *)carve(4 * (size_t)n);
        box_lo = (float4 *)carve(32 * (size_t)n); box_hi = (float4 *)carve(32 * (size_t)n);
        nrm_lo = (float4 *)carve(32 * (size_t)n); nrm_hi = (float4 *)carve(32 * (size_t)n);
        sort_tmp = carve(tmp_bytes);
        HIPCHK(hipMemsetAsync(arrived, 0, 4 * (size_t)n + 4, h->stream));
        uint32_t *top_count = arrived + n;
        const dim3 blk(256), grid((n + 255) / 256);
        if (!sah) {
            f3 flo = {(float)lo[0], (float)lo[1], (float)lo[2]}, inv;
            float ext[3];
            for (int k = 0; k < 3; k++) ext[k] = (float)(hi[k] - lo[k]);
            // cubic Morton cells (one scale for all axes): a flat scene then spends no split on its thin axis
            float emax = std::max(ext[0], std::max(ext[1], ext[2]));
            inv.x = inv.y = inv.z = emax > 0.f ? 1.0f / emax : 0.f;
            // A thin axis the surfaces lie across (a terrain's height) gets the LAST ten bits of the code instead of every
            // third one: a cut on it divides a patch of surface along a contour line where a cut on the other two axes
            // halves it, so the other two are spent first and the thin one only separates what shares their cell
            // (walls, stacked geometry).  HIPEXEC_MORTON_MINOR=-1 keeps the plain interleave, 0..2 names the axis.
            // (HIPEXEC_CLASS_BITS, for both builds: 0 no orientation classes, 1 the three axis classes, 2 normal cells for large
            // triangles: the default, see k_bvh_keys)
            const char *cb = knob("HIPEXEC_CLASS_BITS");
            const int classes = cb ? std::max(0, std::min(2, atoi(cb))) : 2;
            // a triangle counts as large from sqrt(|e1||e2|) = 16 / (scene size) on: from there its loose enlargement, 3e-2 |e1||e2|
            // times the distance, exceeds its own size at scene-size distances
            uint32_t *d_popular = nullptr;
            if (classes == 2) CHK(upload<uint32_t>(h, h->share->popular, sah_detail::CELL_WORDS, &d_popular));
            float big_thr = emax > 0.f ? 256.0f / (emax * emax) : 3.0e38f;
            if (const char *e = knob("HIPEXEC_BIG_SCALE")) big_thr *= (float)atof(e);          // experiment
            int minor = h->morton_minor;
            if (const char *e = knob("HIPEXEC_MORTON_MINOR")) minor = std::max(-1, std::min(2, atoi(e)));
            hipLaunchKernelGGL(k_bvh_keys, grid, blk, 0, h->stream, n, tri_exact, flo, inv, minor, classes, big_thr, d_popular, keys_a);
            HIPCHK(hipGetLastError());
            HIPCHK(hipcub::DeviceRadixSort::SortKeys(sort_tmp, tmp_bytes, keys_a, keys_b, (int)n, 0, 63, h->stream));
        } else {
            // (n >= 2 here: hipexec_optimize leaves smaller scenes alone)
            std::vector<float4> rec;
            try { rec.resize(3 * (size_t)n); } catch (const std::exception &) { return fail(HIPEXEC_E_NOMEM, "host memory for the surface-area build"); }
            HIPCHK(hipMemcpyAsync(rec.data(), tri_exact, rec.size() * sizeof(float4), hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
            HostTopology T;
            int threads = (int)std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
            if (const char *e = knob("HIPEXEC_BUILD_THREADS")) threads = std::max(1, atoi(e));
            try { sah_topology(n, &rec[0].x, REFIT_BLOCK, threads, T, knob("HIPEXEC_CLASS_BITS") ? std::max(0, std::min(2, atoi(knob("HIPEXEC_CLASS_BITS")))) : 2, h->share->popular); }
            catch (const std::exception &e) { return fail(HIPEXEC_E_NOMEM, "surface-area build on the host: %s", e.what()); }     // nothing on the device has been touched yet
            const uint32_t tc = (uint32_t)T.top_list.size();
            HIPCHK(hipMemcpyAsync(keys_b, T.keys.data(), 8 * (size_t)n, hipMemcpyHostToDevice, h->stream));
            HIPCHK(hipMemcpyAsync(child, T.child.data(), 4 * T.child.size(), hipMemcpyHostToDevice, h->stream));
            HIPCHK(hipMemcpyAsync(parent, T.parent.data(), 4 * T.parent.size(), hipMemcpyHostToDevice, h->stream));
            HIPCHK(hipMemcpyAsync(range, T.range.data(), 4 * T.range.size(), hipMemcpyHostToDevice, h->stream));
            if (tc) HIPCHK(hipMemcpyAsync(top_list, T.top_list.data(), 4 * (size_t)tc, hipMemcpyHostToDevice, h->stream));
            HIPCHK(hipMemcpyAsync(top_count, &tc, 4, hipMemcpyHostToDevice, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));        // the host arrays end here
        }
        hipLaunchKernelGGL(k_bvh_leaves, grid, blk, 0, h->stream, n, keys_b, tri_exact, tris, ids, box_lo, box_hi, nrm_lo, nrm_hi);
        HIPCHK(hipGetLastError());
        if (n >= 2) {
            if (!sah) hipLaunchKernelGGL(k_bvh_split, grid, blk, 0, h->stream, n, keys_b, child, parent, range, top_list, top_count);
            HIPCHK(hipGetLastError());
            hipLaunchKernelGGL(k_bvh_refit, dim3((n + REFIT_BLOCK - 1) / REFIT_BLOCK), dim3(REFIT_BLOCK), 0, h->stream, n, child, parent, range, arrived,
                               box_lo, box_hi, nrm_lo, nrm_hi);
            hipLaunchKernelGGL(k_bvh_refit_top, dim3(1), dim3(1024), 0, h->stream, n, child, range, top_list, top_count, arrived, box_lo, box_hi, nrm_lo, nrm_hi);
            HIPCHK(hipGetLastError());
            hipLaunchKernelGGL(k_bvh_emit, grid, blk, 0, h->stream, n, child, range, box_lo, box_hi, nrm_lo, nrm_hi, nodes);
            HIPCHK(hipGetLastError());
        }
        return HIPEXEC_OK;
    }();
    // the kernels above are only queued: the block is released by build_scene after its one synchronisation (or by
    // hipexec_destroy, which synchronises first, if anything fails before that)
    h->build_tmp = block;
    return rc;
}

int build_scene(hipexec *h, const hipexec_scene_desc *s)
{
    DScene &S = h->S;
    S.n_tri = s->n_tri; S.n_sph = s->n_sph; S.n_lamp = s->n_lamp; S.n_mat = s->n_mat;
    S.n_tri_pad = (s->n_tri + TRI_PAD - 1) / TRI_PAD * TRI_PAD;
    if (S.n_tri_pad == 0) S.n_tri_pad = 0;
    h->lamp_words = std::max(1u, (s->n_lamp + 31u) / 32u);
    memcpy(S.cam_pos, s->cam_pos, sizeof S.cam_pos); memcpy(S.cam_mat, s->cam_mat, sizeof S.cam_mat);
    memcpy(S.horizon, s->horizon, sizeof S.horizon); memcpy(S.ambient, s->ambient, sizeof S.ambient);
    S.ambient_factor = s->ambient_factor;

    // centre of the scene's bounding box: origin of the FILTER coordinates
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    bool all_finite = true;
    auto grow = [&](double x, double y, double z) {
        double v[3] = {x, y, z};
        for (int k = 0; k < 3; k++) {
            if (std::isfinite(v[k]) && std::fabs(v[k]) < 1e18) { lo[k] = std::min(lo[k], v[k]); hi[k] = std::max(hi[k], v[k]); }
            else all_finite = false;
        }
    };
    {
        // fast pass in float (the vertices are floats, so their minima and maxima are exact): usable if every coordinate is
        // finite and below 1e18; otherwise the careful pass below, which leaves such coordinates out, does the work
        // (the pass streams 36 bytes per triangle from host memory, which one core does at about 10 GB/s: large scenes
        // are split over a few threads)
        struct Part { float lo[3], hi[3]; int bad; };
        auto scan = [&](size_t t0, size_t t1, Part &pt) {
            float l0 = 3.0e38f, l1 = 3.0e38f, l2 = 3.0e38f, h0 = -3.0e38f, h1 = -3.0e38f, h2 = -3.0e38f;
            int bad = 0;
            const float *P = s->tri_p, *A = s->tri_e1, *Bv = s->tri_e2;
            for (size_t i = 3 * t0; i < 3 * t1; i += 3) {
                const float p0 = P[i], p1 = P[i + 1], p2 = P[i + 2];
                const float a0 = p0 + A[i], a1 = p1 + A[i + 1], a2 = p2 + A[i + 2];
                const float b0 = p0 + Bv[i], b1 = p1 + Bv[i + 1], b2 = p2 + Bv[i + 2];
                l0 = std::min(l0, std::min(p0, std::min(a0, b0))); h0 = std::max(h0, std::max(p0, std::max(a0, b0)));
                l1 = std::min(l1, std::min(p1, std::min(a1, b1))); h1 = std::max(h1, std::max(p1, std::max(a1, b1)));
                l2 = std::min(l2, std::min(p2, std::min(a2, b2))); h2 = std::max(h2, std::max(p2, std::max(a2, b2)));
                // NaN, Inf and huge values fail `< 1e18f`
                bad |= !(std::fabs(p0) < 1e18f) | !(std::fabs(p1) < 1e18f) | !(std::fabs(p2) < 1e18f) | !(std::fabs(a0) < 1e18f) |
                       !(std::fabs(a1) < 1e18f) | !(std::fabs(a2) < 1e18f) | !(std::fabs(b0) < 1e18f) | !(std::fabs(b1) < 1e18f) |
                       !(std::fabs(b2) < 1e18f);
            }
            pt.lo[0] = l0; pt.lo[1] = l1; pt.lo[2] = l2; pt.hi[0] = h0; pt.hi[1] = h1; pt.hi[2] = h2; pt.bad = bad;
        };
        const int n_parts = s->n_tri >= 60000u ? 4 : 1;
        Part parts[4];
        {
            std::vector<std::thread> th;
            for (int q = 1; q < n_parts; q++)
                th.emplace_back(scan, (size_t)s->n_tri * q / n_parts, (size_t)s->n_tri * (q + 1) / n_parts, std::ref(parts[q]));
            scan(0, (size_t)s->n_tri / n_parts, parts[0]);
            for (std::thread &t : th) t.join();
        }
        float flo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, fhi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
        float worst = 0.0f;
        for (int q = 0; q < n_parts; q++) {
            for (int k = 0; k < 3; k++) { flo[k] = std::min(flo[k], parts[q].lo[k]); fhi[k] = std::max(fhi[k], parts[q].hi[k]); }
            if (parts[q].bad) worst = 3.0e38f;
        }
        if (worst < 1e18f) {
            if (s->n_tri) for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], (double)flo[k]); hi[k] = std::max(hi[k], (double)fhi[k]); }
        } else {
            for (uint32_t i = 0; i < s->n_tri; i++) {
                const float *p = s->tri_p + 3 * (size_t)i, *a = s->tri_e1 + 3 * (size_t)i, *b = s->tri_e2 + 3 * (size_t)i;
                grow(p[0], p[1], p[2]); grow(p[0] + a[0], p[1] + a[1], p[2] + a[2]); grow(p[0] + b[0], p[1] + b[1], p[2] + b[2]);
            }
        }
    }
    // bounds of the triangles alone: the grid of the hierarchy's Morton codes (a camera or a sphere far from the mesh would
    // otherwise spend the codes' leading bits on empty space)
    double tri_lo[3], tri_hi[3];
    for (int k = 0; k < 3; k++) { tri_lo[k] = lo[k]; tri_hi[k] = hi[k]; }
    for (uint32_t i = 0; i < s->n_sph; i++) grow(s->sph_center[3 * i], s->sph_center[3 * i + 1], s->sph_center[3 * i + 2]);
    grow(s->cam_pos[0], s->cam_pos[1], s->cam_pos[2]);
    for (int k = 0; k < 3; k++) S.center[k] = (lo[k] <= hi[k]) ? (float)(0.5 * (lo[k] + hi[k])) : 0.f;
    // Radius around the centre that contains every vertex, sphere and the camera (+10 %): ray
    // origins on surfaces lie inside it; the FILTER widens its threshold for farther origins.
    double r2 = 0.0;
    for (int k = 0; k < 3; k++) { double e = (lo[k] <= hi[k]) ? 0.5 * (hi[k] - lo[k]) : 0.0; r2 += e * e; }
    double rs = 0.0;
    for (uint32_t i = 0; i < s->n_sph; i++) if (std::isfinite(s->sph_radius[i])) rs = std::max(rs, (double)std::fabs(s->sph_radius[i]));
    S.rmax = (float)std::max(1e-3, 1.1 * (std::sqrt(r2) + rs));
    {
        const char *nf = knob("HIPEXEC_NO_FILTER");     // verification aid: every pair takes the exact path
        S.theta = (nf && nf[0] == '1') ? 3.0e38f : 1.001f;
    }

    STAGE("host bounds");
    // triangles
    float *d_p = nullptr, *d_e1 = nullptr, *d_e2 = nullptr;
    CHK(upload(h, s->tri_p, 3 * (size_t)s->n_tri, &d_p));
    CHK(upload(h, s->tri_e1, 3 * (size_t)s->n_tri, &d_e1));
    CHK(upload(h, s->tri_e2, 3 * (size_t)s->n_tri, &d_e2));
    h->share->d_tri_p = d_p; h->share->d_tri_e1 = d_e1; h->share->d_tri_e2 = d_e2;
    float4 *ex = nullptr;
    CHK(upload<float4>(h, nullptr, 0, &ex, 3 * (size_t)S.n_tri_pad + 1));
    if (S.n_tri_pad) {
        hipLaunchKernelGGL(k_setup_exact, dim3((S.n_tri_pad + 255) / 256), dim3(256), 0, h->stream, S.n_tri, S.n_tri_pad, d_p, d_e1, d_e2, ex);
        HIPCHK(hipGetLastError());
    }
    S.tri_exact = ex;
    // The records of the index-order scans (filter pairs per camera / arbitrary ray / lamp, chunk boxes) are made by
    // the first call that scans, see ensure_scan_records: the default algorithm never reads them, and an executor that
    // is created per frame (ray_tracer_cl.cpp:14) should not pay for them.
    S.tri_primx = nullptr; S.pair_cam = nullptr; S.pair_gen = nullptr; S.pair_lamp = nullptr; S.chunk_box = nullptr;
    h->share->lamp_pos_host.assign(s->lamp_pos, s->lamp_pos + 3 * (size_t)s->n_lamp);
    {
        const char *nc = knob("HIPEXEC_NO_CULL");       // verification / roofline aid: scan every chunk
        S.cull = (nc && nc[0] == '1') ? 0u : 1u;
    }
    STAGE("triangle upload + exact");
    // Axis for the last bits of the hierarchy's Morton codes: the thinnest one that is under a quarter of the largest extent
    // and that the triangles' normals mostly point along (mean |n_k| > 0.8: surfaces lie across it, as a terrain's height)
    h->morton_minor = -1;
    if (all_finite && s->n_tri) {
        double sum[3] = {0.0, 0.0, 0.0};
        size_t cnt = 0;
        const size_t stride = std::max<size_t>(1, s->n_tri / 65536);           // a sample is enough for a mean
        for (size_t i = 0; i < s->n_tri; i += stride) {
            const float *nn = s->tri_norm + 3 * i;
            if (!(std::fabs(nn[0]) <= 1.5f) || !(std::fabs(nn[1]) <= 1.5f) || !(std::fabs(nn[2]) <= 1.5f)) continue;      // NaN of a zero-area triangle
            for (int k = 0; k < 3; k++) sum[k] += std::fabs(nn[k]);
            cnt++;
        }
        double ext[3], emax = 0.0;
        for (int k = 0; k < 3; k++) { ext[k] = tri_hi[k] - tri_lo[k]; emax = std::max(emax, ext[k]); }
        for (int k = 0; k < 3; k++)
            if (cnt && ext[k] < 0.25 * emax && sum[k] > 0.8 * (double)cnt && (h->morton_minor < 0 || ext[k] < ext[h->morton_minor])) h->morton_minor = k;
    }
    if (all_finite && s->n_tri) sah_detail::popular_cells(s->n_tri, s->tri_e1, s->tri_e2, 3, h->share->popular);
    CHK(build_bvh(h, ex, tri_lo, tri_hi, all_finite));
    STAGE("hierarchy build");
    // what the build does not read goes up on the side stream while the build's kernels run
    h->upload_stream = h->side;
    struct Restore { hipexec *h; ~Restore() { h->upload_stream = nullptr; } } restore{h};
    float *d_norm = nullptr, *d_uv = nullptr; int *d_tmat = nullptr;
    CHK(upload(h, s->tri_norm, 3 * (size_t)s->n_tri, &d_norm));
    CHK(upload(h, s->tri_uv, s->tri_uv ? 6 * (size_t)s->n_tri : 0, &d_uv, 6 * (size_t)std::max(1u, s->n_tri)));
    std::vector<int> tmat(std::max(1u, s->n_tri), (int)s->n_mat);
    for (uint32_t i = 0; i < s->n_tri; i++) tmat[i] = (s->tri_mat && s->tri_mat[i] >= 0) ? s->tri_mat[i] : (int)s->n_mat;
    CHK(upload(h, tmat.data(), tmat.size(), &d_tmat));
    S.tri_norm = d_norm; S.tri_uv = d_uv; S.tri_mat = d_tmat;

    // spheres
    std::vector<float4> sph(std::max(1u, s->n_sph));
    std::vector<int> smat(std::max(1u, s->n_sph), (int)s->n_mat);
    for (uint32_t i = 0; i < s->n_sph; i++) {
        sph[i] = make_float4(s->sph_center[3 * i], s->sph_center[3 * i + 1], s->sph_center[3 * i + 2], s->sph_radius[i]);
        smat[i] = (s->sph_mat && s->sph_mat[i] >= 0) ? s->sph_mat[i] : (int)s->n_mat;
    }
    float4 *d_sph = nullptr; int *d_smat = nullptr;
    CHK(upload(h, sph.data(), sph.size(), &d_sph)); CHK(upload(h, smat.data(), smat.size(), &d_smat));
    S.sph = d_sph; S.sph_mat = d_smat;

    // materials; entry n_mat is the default Material() of scene.h:32-44
    std::vector<float4> ma(s->n_mat + 1), mb(s->n_mat + 1);
    for (uint32_t i = 0; i < s->n_mat; i++) {
        ma[i] = make_float4(s->mat_color[3 * i], s->mat_color[3 * i + 1], s->mat_color[3 * i + 2], s->mat_kd[i]);
        int tex = s->mat_tex ? s->mat_tex[i] : -1;
        float texf; memcpy(&texf, &tex, 4);
        mb[i] = make_float4(s->mat_ks[i], s->mat_hard[i], s->mat_kr[i], texf);
    }
    { int tex = -1; float texf; memcpy(&texf, &tex, 4);
      ma[s->n_mat] = make_float4(0.f, 0.f, 0.f, 0.f); mb[s->n_mat] = make_float4(0.f, 0.f, 0.f, texf); }
    float4 *d_ma = nullptr, *d_mb = nullptr;
    CHK(upload(h, ma.data(), ma.size(), &d_ma)); CHK(upload(h, mb.data(), mb.size(), &d_mb));
    S.mat_a = d_ma; S.mat_b = d_mb;
    // the two Material fields the reference never reads (scene.h:28-29), for the optional refraction mode
    std::vector<float2> mc(s->n_mat + 1, make_float2(0.f, 0.f));
    for (uint32_t i = 0; i < s->n_mat; i++) mc[i] = make_float2(s->mat_kt ? s->mat_kt[i] : 0.f, s->mat_ior ? s->mat_ior[i] : 0.f);
    float2 *d_mc = nullptr;
    CHK(upload(h, mc.data(), mc.size(), &d_mc));
    S.mat_c = d_mc;

    // textures
    uint64_t tex_bytes = 0;
    for (uint32_t i = 0; i < s->n_tex; i++) tex_bytes = std::max<uint64_t>(tex_bytes, s->tex_offset[i] + (uint64_t)s->tex_w[i] * s->tex_h[i] * 4);
    uint8_t *d_tex = nullptr; uint64_t *d_toff = nullptr; uint32_t *d_tw = nullptr, *d_th = nullptr;
    CHK(upload(h, s->tex_rgba, (size_t)tex_bytes, &d_tex, 16));
    CHK(upload(h, s->tex_offset, s->n_tex, &d_toff)); CHK(upload(h, s->tex_w, s->n_tex, &d_tw)); CHK(upload(h, s->tex_h, s->n_tex, &d_th));
    S.tex = d_tex; S.tex_off = d_toff; S.tex_w = d_tw; S.tex_h = d_th;

    // lamps
    std::vector<float4> lamp(std::max(1u, s->n_lamp));
    for (uint32_t i = 0; i < s->n_lamp; i++)
        lamp[i] = make_float4(s->lamp_pos[3 * i], s->lamp_pos[3 * i + 1], s->lamp_pos[3 * i + 2], s->lamp_intensity[i]);
    float4 *d_lamp = nullptr; float *d_ld = nullptr;
    CHK(upload(h, lamp.data(), lamp.size(), &d_lamp)); CHK(upload(h, s->lamp_distance, s->n_lamp, &d_ld));
    S.lamp = d_lamp; S.lamp_dist = d_ld;
    HIPCHK(hipStreamSynchronize(h->side));
    HIPCHK(hipStreamSynchronize(h->stream));
    pool_free(h->device, false, h->build_tmp);          // the hierarchy build has finished with its temporaries
    h->build_tmp = nullptr;
    return HIPEXEC_OK;
}

// ---- launch helpers ---------------------------------------------------------------------
int timed_begin(hipexec *h, int klass, hipStream_t stream)
{
    h->acc.launches[klass]++;
    h->timed_this = ((h->timing >> klass) & 1u) != 0u;
    if (!h->timed_this) return HIPEXEC_OK;
    TimedLaunch t; t.klass = klass;
    for (hipEvent_t *e : {&t.a, &t.b}) {
        if (!h->event_pool.empty()) { *e = h->event_pool.back(); h->event_pool.pop_back(); }
        else HIPCHK(hipEventCreate(e));
    }
    HIPCHK(hipEventRecord(t.a, stream));
    h->timed.push_back(t);
    return HIPEXEC_OK;
}
int timed_end(hipexec *h, hipStream_t stream)
{
    HIPCHK(hipGetLastError());
    if (!h->timed_this) return HIPEXEC_OK;
    HIPCHK(hipEventRecord(h->timed.back().b, stream));
    return HIPEXEC_OK;
}
#define LAUNCH_BS(stream_, klass, kernel, grid, bs, ...)                                       \
    do {                                                                                       \
        CHK(timed_begin(h, klass, stream_));                                                   \
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(bs), 0, stream_, __VA_ARGS__);             \
        CHK(timed_end(h, stream_));                                                            \
    } while (0)
#define LAUNCH_ON(stream_, klass, kernel, grid, ...) LAUNCH_BS(stream_, klass, kernel, grid, 256, __VA_ARGS__)
#define LAUNCH(klass, kernel, grid, ...) LAUNCH_ON(h->stream, klass, kernel, grid, __VA_ARGS__)

// kernel classes to time: none without HIPEXEC_FLAG_TIME_KERNELS, else those named by HIPEXEC_FLAG_TIME_CLASS (all if none is named)
inline uint32_t timing_mask(int32_t flags)
{
    if (!(flags & HIPEXEC_FLAG_TIME_KERNELS)) return 0u;
    uint32_t m = ((uint32_t)flags >> HIPEXEC_FLAG_TIME_CLASS_SHIFT) & ((1u << HIPEXEC_K_COUNT) - 1u);
    return m ? m : (1u << HIPEXEC_K_COUNT) - 1u;
}

inline uint32_t blocks_for(uint64_t threads) { return (uint32_t)((threads + 255) / 256); }

// Records of the index-order scan algorithm, made by the first call that needs them: the camera-independent ones once
// per scene (shared by the executors cloned from one another), the camera-dependent ones per executor and again after
// hipexec_set_camera.
int ensure_scan_records(hipexec *h)
{
    if (h->scan_ready) return HIPEXEC_OK;
    DScene &S = h->S;
    SceneShare *sh = h->share;
    if (!sh) return fail(HIPEXEC_E_INVALID, "a helper lane cannot scan");
    bool make_shared;
    { std::lock_guard<std::mutex> lk(sh->m); make_shared = !sh->scan_shared; }
    // (two executors of one scene must not make their first scanning call at the same time: the calls on the executors
    // of one scene that may allocate scene memory are set-up calls, like hipexec_set_algorithm)
    if (make_shared) {
        PairGen *pg = nullptr;
        CHK(upload<PairGen>(h, nullptr, 0, &pg, (size_t)S.n_tri_pad / 2 + 1));
        if (S.n_tri_pad) {
            f3 cen = {S.center[0], S.center[1], S.center[2]};
            hipLaunchKernelGGL(k_setup_gen_pairs, dim3((S.n_tri_pad / 2 + 255) / 256), dim3(256), 0, h->stream,
                               S.n_tri, S.n_tri_pad, sh->d_tri_p, sh->d_tri_e1, sh->d_tri_e2, cen, S.rmax, pg);
            HIPCHK(hipGetLastError());
        }
        PairCam *pl = nullptr;
        CHK(upload<PairCam>(h, nullptr, 0, &pl, (size_t)S.n_lamp * (S.n_tri_pad / 2) + 1));
        if (S.n_tri_pad) {
            float abs_bound = S.rmax + std::sqrt(S.center[0] * S.center[0] + S.center[1] * S.center[1] + S.center[2] * S.center[2]);
            for (uint32_t l = 0; l < S.n_lamp; l++) {
                f3 L = {sh->lamp_pos_host[3 * l], sh->lamp_pos_host[3 * l + 1], sh->lamp_pos_host[3 * l + 2]};
                hipLaunchKernelGGL(k_setup_lamp_pairs, dim3((S.n_tri_pad / 2 + 255) / 256), dim3(256), 0, h->stream,
                                   S.n_tri, S.n_tri_pad, sh->d_tri_p, sh->d_tri_e1, sh->d_tri_e2, L, abs_bound, pl + (size_t)l * (S.n_tri_pad / 2));
                HIPCHK(hipGetLastError());
            }
        }
        float4 *cb = nullptr;
        uint32_t n_chunks = S.n_tri_pad / CHUNK_TRIS;
        CHK(upload<float4>(h, nullptr, 0, &cb, 3 * (size_t)n_chunks + 1));
        if (n_chunks) {
            hipLaunchKernelGGL(k_setup_chunks, dim3((n_chunks + 63) / 64), dim3(64), 0, h->stream, S.n_tri, n_chunks, sh->d_tri_p, sh->d_tri_e1, sh->d_tri_e2, cb);
            HIPCHK(hipGetLastError());
        }
        HIPCHK(hipStreamSynchronize(h->stream));
        std::lock_guard<std::mutex> lk(sh->m);
        sh->pair_gen = pg; sh->pair_lamp = pl; sh->chunk_box = cb; sh->scan_shared = true;
    }
    { std::lock_guard<std::mutex> lk(sh->m); S.pair_gen = sh->pair_gen; S.pair_lamp = sh->pair_lamp; S.chunk_box = sh->chunk_box; }
    // this executor's camera
    CHK(h->cam_primx.reserve((3 * (size_t)S.n_tri_pad + 1) * sizeof(float4)));
    CHK(h->cam_pairs.reserve(((size_t)S.n_tri_pad / 2 + 1) * sizeof(PairCam)));
    if (S.n_tri_pad) {
        f3 cam = {S.cam_pos[0], S.cam_pos[1], S.cam_pos[2]};
        hipLaunchKernelGGL(k_setup_cam_pairs, dim3((S.n_tri_pad / 2 + 255) / 256), dim3(256), 0, h->stream,
                           S.n_tri, S.n_tri_pad, sh->d_tri_p, sh->d_tri_e1, sh->d_tri_e2, cam, h->cam_primx.as<float4>(), h->cam_pairs.as<PairCam>());
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    S.tri_primx = h->cam_primx.as<float4>(); S.pair_cam = h->cam_pairs.as<PairCam>();
    h->scan_ready = true;
    return HIPEXEC_OK;
}

int call_begin(hipexec *h, uint32_t timing, uint32_t flags)
{
    HIPCHK(hipSetDevice(h->device));
    bool want_bvh = (flags & HIPEXEC_FLAG_SCAN) ? false : ((flags & HIPEXEC_FLAG_BVH) ? true : h->bvh_default);
    h->use_bvh = want_bvh && h->bvh_ok;
    if (!h->use_bvh) CHK(ensure_scan_records(h));
    h->S.bilinear = (flags & HIPEXEC_FLAG_BILINEAR) ? 1u : 0u;
    h->S.refract = (flags & HIPEXEC_FLAG_REFRACT) ? 1u : 0u;
    h->S.ao = (flags & HIPEXEC_FLAG_AO) ? 1u : 0u;
    if (h->S.ao && (!h->use_bvh || h->lamp_words != 1))
        return fail(HIPEXEC_E_LIMIT, "HIPEXEC_FLAG_AO needs the hierarchy algorithm and at most 32 lamps");
    if (h->call_open) {   // a previous asynchronous call has not been collected yet (or failed half-way)
        HIPCHK(hipStreamSynchronize(h->stream));
        if (h->side) HIPCHK(hipStreamSynchronize(h->side));
    }
    // event pairs of a call that was never collected (a failed call, an asynchronous call followed by another call) go
    // back to the pool; they must not be counted as launches of this call
    for (TimedLaunch &t : h->timed) { h->event_pool.push_back(t.a); h->event_pool.push_back(t.b); }
    h->timed.clear();
    h->timing = timing;
    memset(&h->acc, 0, sizeof h->acc);
    h->call_open = true;
    h->lv_slot = 0; h->checks.clear();
    HIPCHK(hipMemsetAsync(h->d_lvl, 0, (size_t)LV_SLOTS * LV_PER * sizeof(uint32_t), h->stream));
    HIPCHK(hipMemsetAsync(h->d_cnt64, 0, CNT64_WORDS * sizeof(unsigned long long), h->stream));
    HIPCHK(hipEventRecord(h->ev_begin, h->stream));
    return HIPEXEC_OK;
}

void add_stats(hipexec_stats &d, const hipexec_stats &a)
{
    d.rays_primary += a.rays_primary; d.rays_bounce += a.rays_bounce; d.rays_shadow += a.rays_shadow;
    d.tests_tri_closest += a.tests_tri_closest; d.tests_sph_closest += a.tests_sph_closest;
    d.tests_tri_any += a.tests_tri_any; d.tests_sph_any += a.tests_sph_any;
    d.tests_tri_primary += a.tests_tri_primary;
    d.bvh_stack_overflows += a.bvh_stack_overflows;
    d.retraced += a.retraced;
    for (int k = 0; k < 3; k++) {
        d.exact_groups[k] += a.exact_groups[k]; d.scanned_chunks[k] += a.scanned_chunks[k];
        d.bvh_nodes[k] += a.bvh_nodes[k]; d.bvh_tests[k] += a.bvh_tests[k];
    }
    for (int k = 0; k < HIPEXEC_K_COUNT; k++) { d.launches[k] += a.launches[k]; d.kernel_ms[k] += a.kernel_ms[k]; }
}

// finish a call: wait, collect timings and counters into *stats (optional)
int call_end(hipexec *h, hipexec_stats *stats)
{
    HIPCHK(hipEventRecord(h->ev_end, h->stream));
    HIPCHK(hipMemcpyAsync(h->h_cnt64, h->d_cnt64, CNT64_WORDS * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    for (uint32_t st = 1; st < CNT_STRIPES; st++)           // the stripes add up in stripe 0
        for (int k = 0; k < N_CNT64; k++) h->h_cnt64[k] += h->h_cnt64[(size_t)st * CNT_STRIDE + k];
    h->acc.tests_tri_any = h->h_cnt64[0];
    h->acc.tests_sph_any = h->h_cnt64[1];
    for (int k = 0; k < 3; k++) {
        h->acc.exact_groups[k] = h->h_cnt64[2 + k]; h->acc.scanned_chunks[k] = h->h_cnt64[5 + k];
        h->acc.bvh_nodes[k] = h->h_cnt64[8 + k]; h->acc.bvh_tests[k] = h->h_cnt64[11 + k];
    }
    h->acc.rays_shadow += h->h_cnt64[14];
    h->acc.bvh_stack_overflows = h->h_cnt64[15];
    // bounce rays generated inside k_bvh_tail were never counted on the host
    h->acc.rays_bounce += h->h_cnt64[16];
    h->acc.tests_tri_closest += h->h_cnt64[16] * (uint64_t)h->S.n_tri;
    h->acc.tests_sph_closest += h->h_cnt64[16] * (uint64_t)h->S.n_sph;
    h->acc.algorithm = h->use_bvh ? HIPEXEC_ALGO_BVH : HIPEXEC_ALGO_SCAN;
    h->acc.lanes = (uint32_t)h->lanes_used;
    HIPCHK(hipEventElapsedTime(&h->acc.total_ms, h->ev_begin, h->ev_end));
    for (TimedLaunch &t : h->timed) {
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, t.a, t.b));
        h->acc.kernel_ms[t.klass] += ms;
        h->event_pool.push_back(t.a); h->event_pool.push_back(t.b);
    }
    h->timed.clear();
    h->call_open = false;
    // Batches whose launches were sized on the host: a level that produced more rays than its launch covered has lost
    // rays, and the frame must be traced again; otherwise the sizes the kernels produced are the frame's ray counts and
    // what the next frame with this key is sized from.
    for (const LevelCheck &c : h->checks) {
        const uint32_t *got = h->h_lvl + (size_t)c.slot * LV_PER;
        bool covered = true;
        int top = 0;
        for (int L = 1; L < LV_PER; L++) {
            const uint32_t launched = (size_t)L < c.launched.size() ? c.launched[(size_t)L] : 0u;
            covered = covered && got[L] <= launched;
            if (got[L]) top = L;
        }
        if (!covered) {
            h->redo = true;
            if (knob("HIPEXEC_TRACE_REDO"))                  // which level outgrew its launch, and by how much
                for (int L = 1; L < LV_PER; L++) {
                    const uint32_t launched = (size_t)L < c.launched.size() ? c.launched[(size_t)L] : 0u;
                    if (got[L] || launched) fprintf(stderr, "hipexec: level %d launched for %u rays, queue held %u%s\n", L, launched, got[L], got[L] > launched ? "  <-- too small" : "");
                }
            continue;
        }
        for (int L = 1; L <= top; L++) {
            h->acc.rays_bounce += got[L];
            h->acc.tests_tri_closest += (uint64_t)got[L] * h->S.n_tri;
            h->acc.tests_sph_closest += (uint64_t)got[L] * h->S.n_sph;
        }
        const int seen = c.tail_level >= 0 ? c.tail_level : LV_PER - 1;
        LevelPred rec; rec.key = c.key; rec.seen = seen; rec.n.assign(got, got + top + 1); rec.n[0] = 0u;
        // test hook: a remembered size that is wrong, so that the check above and the repeat of the frame can be exercised
        if (h->predict_spoil && rec.n.size() > 1 && (int64_t)rec.n[1] + h->predict_spoil > 0)
            rec.n[1] = (uint32_t)((int64_t)rec.n[1] + h->predict_spoil);
        std::lock_guard<std::mutex> lk(h->pred_home->pred_m);
        std::vector<LevelPred> &preds = h->pred_home->preds;
        for (size_t i = 0; i < preds.size(); i++)
            if (preds[i].key == c.key) {
                const LevelPred &old = preds[i];
                auto at = [](const std::vector<uint32_t> &v, int L) { return (size_t)L < v.size() ? v[(size_t)L] : 0u; };
                rec.steady = true;
                rec.delta = old.delta;
                rec.delta.resize(std::max(rec.n.size(), old.n.size()), 0u);
                for (int L = 1; L <= std::min(seen, old.seen); L++) {
                    const uint32_t a = at(old.n, L), b = at(rec.n, L);
                    rec.steady = rec.steady && a == b;
                    if ((size_t)L < rec.delta.size()) rec.delta[(size_t)L] = a > b ? a - b : b - a;
                }
                if (old.n.size() > (size_t)seen + 1) {        // levels this frame left to the tail kernel: what is known of them stays
                    rec.n.resize((size_t)seen + 1, 0u);
                    rec.n.insert(rec.n.end(), old.n.begin() + seen + 1, old.n.end());
                }
                preds.erase(preds.begin() + (long)i);
                break;
            }
        if (preds.size() >= 64) preds.erase(preds.begin());
        preds.push_back(rec);
    }
    h->checks.clear();
    // the other lanes' shares of the frame: their counters and kernel times are added to this call's
    for (hipexec *g : h->helpers) {
        if (!g->call_open) continue;
        CHK(call_end(g, nullptr));
        add_stats(h->acc, g->acc);
        if (g->redo) { h->redo = true; g->redo = false; }
    }
    h->lanes_used = 1;
    h->acc.retraced = h->retraced;
    if (stats) *stats = h->acc;
    return HIPEXEC_OK;
}

// Queue sizes of the recursion levels of the batch being traced (see LevelPred).
struct LevelRun {
    int slot = -1;                       // counters of this batch: d_lvl[slot][level]; -1: the call has more batches than slots
    bool sized = false;                  // launch sizes are chosen on the host (else: read back from the device, level by level)
    bool have_pred = false;              // remembered sizes exist (sized runs)
    LevelPred pred;
    std::array<int64_t, 24> key{};
    std::vector<uint32_t> launched;      // launched[L]: size the launches of level L were made for (sized runs)
    int tail_level = -1;                 // level from which k_bvh_tail took the batch over (-1: it did not)
};

int level_begin(hipexec *h, const std::array<int64_t, 24> &key, bool may_size, LevelRun &run)
{
    run.slot = h->lv_slot < LV_SLOTS ? h->lv_slot++ : -1;
    run.have_pred = false;
    run.key = key; run.launched.assign(1, 0u);
    run.sized = run.slot >= 0 && may_size && h->size_mode != SIZE_WAIT;
    if (run.sized && h->size_mode == SIZE_AUTO) {
        std::lock_guard<std::mutex> lk(h->pred_home->pred_m);
        for (const LevelPred &p : h->pred_home->preds) if (p.key == key) { run.pred = p; run.have_pred = true; }
    }
    return HIPEXEC_OK;
}

// device counter that the producers of level `level`'s rays add to
uint32_t *level_counter(hipexec *h, const LevelRun &run, int level)
{
    return run.slot >= 0 ? h->d_lvl + (size_t)run.slot * LV_PER + level : h->d_cnt32;
}

// what the consumers of level `level`'s queue get as its device-side fill: the counter when the launch size was chosen
// on the host, nothing when it was read back from that counter (or is level 0, whose size the host computes)
const uint32_t *level_fill(hipexec *h, const LevelRun &run, int level)
{
    return (run.sized && level >= 1) ? level_counter(h, run, level) : nullptr;
}

// Size for the launches of `level` (>= 1), once its producers are enqueued on `stream`; `prev` is the size the level
// before was launched with, which bounds it.  `counted`: the value is the true count (the caller adds it to the ray
// statistics; sized runs get theirs from the device counters when the call is collected).
int level_count(hipexec *h, LevelRun &run, int level, uint32_t prev, uint32_t *out, bool *counted)
{
    *counted = !run.sized;
    if (run.sized) {
        uint32_t n = prev;                                              // the bound: at most one successor per ray
        // (a level the remembered frame did not count itself is known only from an older frame's entry, and that only
        // while the counted levels repeat; otherwise nothing is known of it and the bound stands)
        if (run.have_pred && (level <= run.pred.seen || (run.pred.steady && (size_t)level < run.pred.n.size()))) {
            const uint32_t last = (size_t)level < run.pred.n.size() ? run.pred.n[(size_t)level] : 0u;
            // a quarter more than the last frame produced, twice the last change (a moving camera changes a level's queue
            // by about as much from frame to frame; the deep levels of a depth-5 frame grew by 18 % over the three frames
            // a loop with three frames in flight looks back) and a few blocks' worth of slack.  A launch that is too
            // large costs blocks that exit at once (0.5 ns each), so a frame that repeats is sized the same way: the first
            // frame after a camera starts to move then still fits.
            const uint32_t moved = (size_t)level < run.pred.delta.size() ? run.pred.delta[(size_t)level] : 0u;
            const uint64_t want = (uint64_t)last + last / 4u + 2ull * moved + 4096u;
            n = (uint32_t)std::min<uint64_t>(prev, want);
        }
        run.launched.resize((size_t)level + 1, 0u);
        run.launched[(size_t)level] = n;
        *out = n;
        return HIPEXEC_OK;
    }
    uint32_t *src = level_counter(h, run, level);
    HIPCHK(hipMemcpyAsync(h->h_cnt32, src, sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    *out = h->h_cnt32[0];
    return HIPEXEC_OK;
}

int level_end(hipexec *h, LevelRun &run)
{
    if (run.sized) {
        // the counts the kernels produced come back with the call and are compared with the launches then (call_end)
        HIPCHK(hipMemcpyAsync(h->h_lvl + (size_t)run.slot * LV_PER, h->d_lvl + (size_t)run.slot * LV_PER, LV_PER * sizeof(uint32_t),
                              hipMemcpyDeviceToHost, h->stream));
        h->checks.push_back(LevelCheck{run.slot, run.key, run.launched, run.tail_level});
    }
    return HIPEXEC_OK;
}

template <int MODE>
int launch_anyhit(hipexec *h, const float4 *rays, uint32_t n, uint32_t *occl, uint32_t lw)
{
    if (h->use_bvh) {
        LAUNCH(HIPEXEC_K_ANYHIT, (k_bvh_anyhit<MODE>), blocks_for(n), h->S, h->B, rays, n, occl, lw, h->d_cnt64);
        return HIPEXEC_OK;
    }
    constexpr int R = 2;
    uint64_t waves = ((uint64_t)n + 64 * R - 1) / (64 * R);
    LAUNCH(HIPEXEC_K_ANYHIT, (k_anyhit<MODE, R>), blocks_for(waves * 64), h->S, rays, n, occl, lw, h->d_cnt64);
    return HIPEXEC_OK;
}

template <int MODE, int R>
int launch_anyhit_lamp_r(hipexec *h, uint32_t lamp, const float4 *rays, uint32_t n, uint32_t *occl, uint32_t lw)
{
    uint64_t waves = ((uint64_t)n + 64 * R - 1) / (64 * R);
    const PairCam *pl = h->S.pair_lamp + (size_t)lamp * (h->S.n_tri_pad / 2);
    if (h->anyhit_lds)
        LAUNCH(HIPEXEC_K_ANYHIT, (k_anyhit_lamp<MODE, R, true>), blocks_for(waves * 64), h->S, pl, rays, n, occl, lw, h->d_cnt64);
    else
        LAUNCH(HIPEXEC_K_ANYHIT, (k_anyhit_lamp<MODE, R, false>), blocks_for(waves * 64), h->S, pl, rays, n, occl, lw, h->d_cnt64);
    return HIPEXEC_OK;
}
template <int MODE>
int launch_anyhit_lamp(hipexec *h, uint32_t lamp, const float4 *rays, uint32_t n, uint32_t *occl, uint32_t lw)
{
    if (h->use_bvh) return launch_anyhit<MODE>(h, rays, n, occl, lw);
    switch (h->anyhit_r) {
    case 1: return launch_anyhit_lamp_r<MODE, 1>(h, lamp, rays, n, occl, lw);
    case 4: return launch_anyhit_lamp_r<MODE, 4>(h, lamp, rays, n, occl, lw);
    default: return launch_anyhit_lamp_r<MODE, 2>(h, lamp, rays, n, occl, lw);
    }
}

// Rays owned by one block of a traversal kernel: `full` when the launch has work for the whole chip,
// fewer (in whole waves) for a small launch, so that it still spreads over all compute units
// instead of running a few long blocks on some of them.
static uint32_t small_launch_share(uint32_t n, uint32_t full)
{
    const uint32_t want_blocks = 2048;                       // 8 per compute unit
    uint32_t share = ((n / want_blocks) + 63u) & ~63u;
    return std::max(64u, std::min(full, share));
}

template <int MODE>
int launch_bvh_closest(hipexec *h, int klass_time, const float4 *rays, uint32_t n, float4 *hits, int klass_count,
                       const uint32_t *n_dev = nullptr)
{
    if (h->packet && klass_count == 0) {
        // camera rays: the 64 rays of a wave are one 8x8-pixel tile of one sub-sample and walk the hierarchy as a packet
        LAUNCH(klass_time, (k_bvh_closest_pk<MODE, false>), blocks_for(n), h->S, h->B, const_cast<float4 *>(rays), n, hits, h->d_cnt64,
               klass_count, Frame{});
    } else if (h->bvh_refill && n < h->simple_from) {
        // (k_bvh_closest_p: a block owns per_block rays and its lanes take them one at a time.  It was 12 % ahead of one ray per
        // lane with binary nodes; with 4-wide nodes and 6 waves per SIMD for both, one ray per lane is 2-5 % ahead at every
        // launch size measured, and is what runs unless HIPEXEC_SIMPLE_FROM says otherwise.)
        uint32_t per_block = h->closest_per_block ? (uint32_t)h->closest_per_block
                                                  : small_launch_share(n, klass_count == 0 ? 256u : 768u);
        uint32_t blocks = (n + per_block - 1) / per_block;
        LAUNCH(klass_time, (k_bvh_closest_p<MODE>), blocks, h->S, h->B, rays, n, hits, h->d_cnt64, klass_count,
               h->refill_closest | (h->sched_w << 8) | (int)(per_block << 16), n_dev);
    } else {
        // the node array once more as a restrict-qualified argument: steps in which every lane of a wave sits on the same
        // node (the top of the tree for any rays that start together) fetch the record through the scalar cache and leave
        // the texture addressers, 74 % busy in this kernel, to the steps whose lanes have spread out
        const float4 *closest_sc = (h->uniform_fetch & 2u) ? h->B.nodes : nullptr;
        if (h->block_closest == 64 || n <= h->block64_below)
            LAUNCH_BS(h->stream, klass_time, (k_bvh_closest<MODE, 64>), (n + 63u) / 64u, 64, h->S, h->B, rays, n, hits, h->d_cnt64, klass_count, n_dev, closest_sc);
        else if (h->block_closest == 128)
            LAUNCH_BS(h->stream, klass_time, (k_bvh_closest<MODE, 128>), (n + 127u) / 128u, 128, h->S, h->B, rays, n, hits, h->d_cnt64, klass_count, n_dev, closest_sc);
        else
            LAUNCH(klass_time, (k_bvh_closest<MODE, 256>), blocks_for(n), h->S, h->B, rays, n, hits, h->d_cnt64, klass_count, n_dev, closest_sc);
    }
    return HIPEXEC_OK;
}

template <int MODE, int R>
int launch_closest_r(hipexec *h, const float4 *rays, uint32_t n, float4 *hits)
{
    uint64_t waves = ((uint64_t)n + 64 * R - 1) / (64 * R);
    if (h->closest_lds)
        LAUNCH(HIPEXEC_K_CLOSEST, (k_closest<MODE, R, true>), blocks_for(waves * 64), h->S, rays, n, hits, h->d_cnt64);
    else
        LAUNCH(HIPEXEC_K_CLOSEST, (k_closest<MODE, R, false>), blocks_for(waves * 64), h->S, rays, n, hits, h->d_cnt64);
    return HIPEXEC_OK;
}
template <int MODE>
int launch_closest(hipexec *h, const float4 *rays, uint32_t n, float4 *hits, const uint32_t *n_dev = nullptr)
{
    if (h->use_bvh) return launch_bvh_closest<MODE>(h, HIPEXEC_K_CLOSEST, rays, n, hits, 1, n_dev);
    switch (h->closest_r) {
    case 1: return launch_closest_r<MODE, 1>(h, rays, n, hits);
    case 3: return launch_closest_r<MODE, 3>(h, rays, n, hits);
    case 4: return launch_closest_r<MODE, 4>(h, rays, n, hits);
    default: return launch_closest_r<MODE, 2>(h, rays, n, hits);
    }
}

constexpr uint32_t OVERLAP_MAX_SAMPLES = 2500000u;

// Shading inside k_bvh_shadow (the occlusion bits of a block stay in LDS): needs them to fit one word per sample.
// Measured: worth 2-6 % of a whole frame with few lamps (C2, C3, C4); with 16 lamps the shading loop in the same kernel
// costs the traversal registers (C5 +1 %), and on the side stream of a small batch a separate k_shade overlaps better
// (1/8 of C2 +5 %), so those keep the two launches.
bool fused_shading(const hipexec *h, bool side_stream)
{
    if (h->S.ao) return true;                 // the occlusion sum of a sample stays in the registers of the kernel that traced it
    if (h->lamp_words != 1 || h->fuse_shade == 0) return false;
    return h->fuse_shade == 2 || (!side_stream && h->S.n_lamp <= 8);
}

// k_bvh_tail takes over when the queue of a level is small and at least two levels remain (for the last
// level alone the separate, coherent kernels are as good).
bool use_tail(const hipexec *h, uint32_t n_rays, int level, int depth)
{
    return h->use_bvh && h->bvh_refill && n_rays <= h->tail_max && level < depth && !h->S.ao;
}

template <int MODE>
int launch_tail(hipexec *h, const float4 *rays, uint32_t n, int level, int depth, size_t stride, const uint32_t *n_dev)
{
    if (h->block_tail == 64 || n <= h->block64_below)
        LAUNCH_BS(h->stream, HIPEXEC_K_CLOSEST, (k_bvh_tail<MODE, 64>), (n + 63u) / 64u, 64, h->S, h->B, rays, n, level, depth,
                  h->local.as<float4>(), stride, h->d_cnt64, n_dev);
    else
        LAUNCH(HIPEXEC_K_CLOSEST, (k_bvh_tail<MODE, 256>), blocks_for(n), h->S, h->B, rays, n, level, depth, h->local.as<float4>(), stride,
               h->d_cnt64, n_dev);
    return HIPEXEC_OK;
}

// Shadow rays of a level (FUSED: and its shading) on `stream`.  Two forms of the kernel: the direct one (a lane keeps its
// sample and goes through the lamps) and the one that compacts the block's (sample, lamp) pairs into a list in LDS first.
// Measured on whole frames: equal on C2, C3 and C4, the direct form 9 % faster on C5 (16 lamps: four rounds of
// compaction with their barriers); HIPEXEC_SHADOW_DIRECT=0 selects the list form.
template <int MODE, bool FUSED>
int launch_bvh_shadow(hipexec *h, hipStream_t stream, const float4 *rays, const float4 *hits, uint32_t n, float4 *local_level, bool spawn,
                      float4 *next_rays, uint32_t *next_count, const uint32_t *n_dev)
{
    const uint32_t per_block = small_launch_share(n, h->shadow_per_block);
    const uint32_t blocks = (n + per_block - 1) / per_block;
    const float4 *nodes_sc = (h->uniform_fetch & 1u) ? h->B.nodes : nullptr;
    const bool direct = h->shadow_direct || h->S.ao;       // the occlusion rays of HIPEXEC_FLAG_AO exist in the direct form only
    const bool one_wave = h->block_shadow == 64 || n <= h->block64_below;
    if (direct && h->S.ao && FUSED && MODE == HIPEXEC_MODE_CPU) {
        if (one_wave)
            LAUNCH_BS(stream, HIPEXEC_K_ANYHIT, (k_bvh_shadow_direct<MODE, FUSED, 64, true>), (n + 63u) / 64u, 64, h->S, h->B, rays, hits, n,
                      h->occl.as<uint32_t>(), h->lamp_words, h->d_cnt64, 64u, nodes_sc, local_level, (int)spawn, next_rays, next_count, n_dev);
        else
            LAUNCH_ON(stream, HIPEXEC_K_ANYHIT, (k_bvh_shadow_direct<MODE, FUSED, 256, true>), blocks, h->S, h->B, rays, hits, n, h->occl.as<uint32_t>(),
                      h->lamp_words, h->d_cnt64, per_block, nodes_sc, local_level, (int)spawn, next_rays, next_count, n_dev);
    } else if (direct && h->S.n_lamp >= h->shadow_waves6_from) {          // the variant compiled for RT_SHADOW_WAVES_MANY waves per SIMD
        if (one_wave)
            LAUNCH_BS(stream, HIPEXEC_K_ANYHIT, (k_bvh_shadow_direct<MODE, FUSED, 64, false, RT_SHADOW_WAVES_MANY>), (n + 63u) / 64u, 64, h->S, h->B, rays, hits, n,
                      h->occl.as<uint32_t>(), h->lamp_words, h->d_cnt64, 64u, nodes_sc, local_level, (int)spawn, next_rays, next_count, n_dev);
        else
            LAUNCH_ON(stream, HIPEXEC_K_ANYHIT, (k_bvh_shadow_direct<MODE, FUSED, 256, false, RT_SHADOW_WAVES_MANY>), blocks, h->S, h->B, rays, hits, n, h->occl.as<uint32_t>(),
                      h->lamp_words, h->d_cnt64, per_block, nodes_sc, local_level, (int)spawn, next_rays, next_count, n_dev);
    } else if (direct && one_wave)
        LAUNCH_BS(stream, HIPEXEC_K_ANYHIT, (k_bvh_shadow_direct<MODE, FUSED, 64>), (n + 63u) / 64u, 64, h->S, h->B, rays, hits, n,
                  h->occl.as<uint32_t>(), h->lamp_words, h->d_cnt64, 64u, nodes_sc, local_level, (int)spawn, next_rays, next_count, n_dev);
    else if (direct)
        LAUNCH_ON(stream, HIPEXEC_K_ANYHIT, (k_bvh_shadow_direct<MODE, FUSED, 256>), blocks, h->S, h->B, rays, hits, n, h->occl.as<uint32_t>(),
                  h->lamp_words, h->d_cnt64, per_block, nodes_sc, local_level, (int)spawn, next_rays, next_count, n_dev);
    else
        LAUNCH_ON(stream, HIPEXEC_K_ANYHIT, (k_bvh_shadow<MODE, FUSED>), blocks, h->S, h->B, rays, hits, n, h->occl.as<uint32_t>(),
                  h->lamp_words, h->d_cnt64, per_block, nodes_sc, local_level, (int)spawn, next_rays, next_count, n_dev);
    return HIPEXEC_OK;
}

// trace_levels for the hierarchy with reflections, on two streams.  The bounce rays of a level depend
// only on its hits, so k_bounce queues them at once and the main stream goes on to trace the next
// level, while the side stream casts this level's shadow rays and shades it.  The small, long-tailed
// launches of the deep levels then run beside each other instead of one after the other.
// Buffers alternate by level; a level's ray and hit buffers are reused two levels later, which the
// main stream does only after the side stream has finished with them (ev_lit).
template <int MODE>
int trace_levels_overlapped(hipexec *h, uint32_t n0, size_t stride, int depth, int *levels_used, LevelRun &run)
{
    const DScene &S = h->S;
    float4 *ray_buf[2] = {h->rayA.as<float4>(), h->rayB.as<float4>()};
    float4 *hit_buf[2] = {h->hitM.as<float4>(), h->hitN.as<float4>()};
    float4 *cur_rays = ray_buf[0], *cur_hits = h->hit0.as<float4>();
    hipStream_t main_s = h->stream, side_s = h->side;
    uint32_t n = n0;
    int level = 0;
    bool tail = false;
    for (;; level++) {
        const bool spawn = level < depth;
        const uint32_t *n_dev = level_fill(h, run, level);
        HIPCHK(hipEventRecord(h->ev_hits[level & 1], main_s));               // this level's hits are complete
        if (spawn) {
            if (level >= 1) HIPCHK(hipStreamWaitEvent(main_s, h->ev_lit[(level - 1) & 1], 0));
            if (run.slot < 0) HIPCHK(hipMemsetAsync(h->d_cnt32, 0, sizeof(uint32_t), main_s));
            LAUNCH_ON(main_s, HIPEXEC_K_SHADE, (k_bounce<MODE>), blocks_for(n), S, cur_rays, cur_hits, n, ray_buf[(level + 1) & 1],
                      level_counter(h, run, level + 1), n_dev);
        }
        HIPCHK(hipStreamWaitEvent(side_s, h->ev_hits[level & 1], 0));
        float4 *local_level = h->local.as<float4>() + (size_t)level * stride;
        if (fused_shading(h, true)) {
            CHK((launch_bvh_shadow<MODE, true>(h, side_s, cur_rays, cur_hits, n, local_level, spawn, nullptr, nullptr, n_dev)));
        } else {
            if (S.n_lamp) CHK((launch_bvh_shadow<MODE, false>(h, side_s, cur_rays, cur_hits, n, nullptr, false, nullptr, nullptr, n_dev)));
            LAUNCH_ON(side_s, HIPEXEC_K_SHADE, (k_shade<MODE>), blocks_for(n), S, cur_rays, cur_hits, n, h->occl.as<uint32_t>(),
                      h->lamp_words, local_level, spawn, (float4 *)nullptr, (uint32_t *)nullptr, n_dev);
        }
        HIPCHK(hipEventRecord(h->ev_lit[level & 1], side_s));
        if (!spawn) break;
        uint32_t nn = 0; bool counted = false;
        CHK(level_count(h, run, level + 1, n, &nn, &counted));
        if (nn == 0) break;
        if (counted) {
            h->acc.rays_bounce += nn;
            h->acc.tests_tri_closest += (uint64_t)nn * S.n_tri;
            h->acc.tests_sph_closest += (uint64_t)nn * S.n_sph;
        }
        cur_rays = ray_buf[(level + 1) & 1];
        cur_hits = hit_buf[(level + 1) & 1];
        if (use_tail(h, nn, level + 1, depth)) {
            CHK(launch_tail<MODE>(h, cur_rays, nn, level + 1, depth, stride, level_fill(h, run, level + 1)));
            run.tail_level = level + 1;
            tail = true;
            break;
        }
        CHK(launch_bvh_closest<MODE>(h, HIPEXEC_K_CLOSEST, cur_rays, nn, cur_hits, 1, level_fill(h, run, level + 1)));
        n = nn;
    }
    HIPCHK(hipStreamWaitEvent(main_s, h->ev_lit[level & 1], 0));             // the side stream is in order: every level is lit
    *levels_used = tail ? depth + 1 : level + 1;
    return HIPEXEC_OK;
}

template <int MODE>
int trace_levels_run(hipexec *h, uint32_t n0, size_t stride, int max_depth, int *levels_used, LevelRun &run);

// Shadow rays, shading and bounces for n0 level-0 rays whose records are in rayA and whose
// closest hits are in hit0.  Leaves per-level colours in h->local ([level][stride]).
template <int MODE>
int trace_levels(hipexec *h, uint32_t n0, size_t stride, int max_depth, int *levels_used, const std::array<int64_t, 24> &key)
{
    LevelRun run;
    CHK(level_begin(h, key, h->use_bvh, run));
    int rc = trace_levels_run<MODE>(h, n0, stride, max_depth, levels_used, run);
    if (rc == HIPEXEC_OK) rc = level_end(h, run);
    return rc;
}

template <int MODE>
int trace_levels_run(hipexec *h, uint32_t n0, size_t stride, int max_depth, int *levels_used, LevelRun &run)
{
    const DScene &S = h->S;
    // Measured on C2 / C3 shards: with up to ~2 M samples per lane the overlap wins (1/8 of C2: 1.54 -> 1.45 ms,
    // 1/8 of C3: 3.47 -> 3.36 ms); larger batches fill the chip from the two lanes alone and only pay for k_bounce.
    const bool overlap = h->overlap_levels == 2 || (h->overlap_levels == 1 && n0 <= OVERLAP_MAX_SAMPLES);
    if (MODE == HIPEXEC_MODE_CPU && max_depth > 0 && h->use_bvh && h->bvh_refill && h->side && overlap)
        return trace_levels_overlapped<MODE>(h, n0, stride, max_depth, levels_used, run);
    float4 *cur_rays = h->rayA.as<float4>(), *next_rays = h->rayB.as<float4>();
    float4 *cur_hits = h->hit0.as<float4>();
    uint32_t n = n0;
    int depth = (MODE == HIPEXEC_MODE_CL) ? 0 : std::max(0, max_depth);
    int level = 0;
    for (;; level++) {
        if (!h->use_bvh || run.slot < 0) HIPCHK(hipMemsetAsync(h->d_cnt32, 0, h->n_cnt32 * sizeof(uint32_t), h->stream));
        const bool spawn = level < depth;
        float4 *local_level = h->local.as<float4>() + (size_t)level * stride;
        uint32_t *next_count = level_counter(h, run, level + 1);
        const uint32_t *n_dev = level_fill(h, run, level);
        bool shaded = false;
        if (h->use_bvh && fused_shading(h, false)) {
            CHK((launch_bvh_shadow<MODE, true>(h, h->stream, cur_rays, cur_hits, n, local_level, spawn, next_rays, next_count, n_dev)));
            shaded = true;
        } else if (S.n_lamp && h->use_bvh) {
            CHK((launch_bvh_shadow<MODE, false>(h, h->stream, cur_rays, cur_hits, n, nullptr, false, nullptr, nullptr, n_dev)));
        } else if (S.n_lamp) {
            LAUNCH(HIPEXEC_K_SHADE, (k_shadowgen<MODE>), blocks_for(n), S, cur_rays, cur_hits, n,
                   h->shadowq.as<float4>(), h->d_cnt32 + 1);
            HIPCHK(hipMemcpyAsync(h->h_cnt32 + 1, h->d_cnt32 + 1, S.n_lamp * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
            HIPCHK(hipStreamSynchronize(h->stream));
            for (uint32_t l = 0; l < S.n_lamp; l++) {
                uint32_t nsh = h->h_cnt32[1 + l];
                h->acc.rays_shadow += nsh;
                if (nsh) CHK(launch_anyhit_lamp<MODE>(h, l, h->shadowq.as<float4>() + 2 * (size_t)l * n, nsh,
                                                      h->occl.as<uint32_t>(), h->lamp_words));
            }
        }
        if (!shaded)
            LAUNCH(HIPEXEC_K_SHADE, (k_shade<MODE>), blocks_for(n), S, cur_rays, cur_hits, n, h->occl.as<uint32_t>(),
                   h->lamp_words, local_level, spawn, next_rays, next_count, n_dev);
        if (!spawn) break;
        uint32_t nn = 0; bool counted = false;
        CHK(level_count(h, run, level + 1, n, &nn, &counted));
        if (nn == 0) break;
        if (counted) {
            h->acc.rays_bounce += nn;
            h->acc.tests_tri_closest += (uint64_t)nn * S.n_tri;
            h->acc.tests_sph_closest += (uint64_t)nn * S.n_sph;
        }
        if (use_tail(h, nn, level + 1, depth)) {
            CHK(launch_tail<MODE>(h, next_rays, nn, level + 1, depth, stride, level_fill(h, run, level + 1)));
            run.tail_level = level + 1;
            *levels_used = depth + 1;
            return HIPEXEC_OK;
        }
        CHK(launch_closest<MODE>(h, next_rays, nn, h->hitN.as<float4>(), level_fill(h, run, level + 1)));
        std::swap(cur_rays, next_rays);
        cur_hits = h->hitN.as<float4>();
        n = nn;
    }
    *levels_used = level + 1;
    return HIPEXEC_OK;
}

int reserve_trace_buffers(hipexec *h, uint64_t n0, int levels)
{
    CHK(h->rayA.reserve(n0 * 32)); CHK(h->rayB.reserve(n0 * 32));
    CHK(h->hit0.reserve(n0 * 16)); CHK(h->hitN.reserve(n0 * 16));
    if (levels > 1 && h->use_bvh && h->side && h->overlap_levels) CHK(h->hitM.reserve(n0 * 16));
    CHK(h->local.reserve(n0 * 16 * (uint64_t)levels));
    size_t occl_bytes = n0 * 4 * h->lamp_words;
    if (occl_bytes > h->occl.cap) {
        CHK(h->occl.reserve(occl_bytes));
        HIPCHK(hipMemsetAsync(h->occl.p, 0, h->occl.cap, h->stream));
    }
    if (!h->use_bvh) CHK(h->shadowq.reserve(std::max<uint64_t>(1, n0 * h->S.n_lamp) * 32));
    return HIPEXEC_OK;
}

int n_tiles_of(const hipexec_render_cfg *c)
{
    int total = (c->height + HIPEXEC_TILE_ROWS - 1) / HIPEXEC_TILE_ROWS;
    if (c->tile_first >= total) return 0;
    return (total - c->tile_first + c->tile_step - 1) / c->tile_step;
}

int validate_cfg(const hipexec_render_cfg *c)
{
    if (!c) return fail(HIPEXEC_E_INVALID, "render cfg is NULL");
    if (c->width <= 0 || c->height <= 0) return fail(HIPEXEC_E_INVALID, "frame size %dx%d is not positive", c->width, c->height);
    if (c->width > 65536 || c->height > 65536) return fail(HIPEXEC_E_LIMIT, "frame size %dx%d exceeds 65536", c->width, c->height);
    if (c->mode != HIPEXEC_MODE_CPU && c->mode != HIPEXEC_MODE_CL) return fail(HIPEXEC_E_INVALID, "unknown mode %d", c->mode);
    if (c->max_depth < 0 || c->max_depth > 64) return fail(HIPEXEC_E_LIMIT, "max_depth %d outside [0,64]", c->max_depth);
    if (c->tile_first < 0 || c->tile_step < 1) return fail(HIPEXEC_E_INVALID, "bad tile selection %d/%d", c->tile_first, c->tile_step);
    return HIPEXEC_OK;
}

// Renders the tiles sub, sub + n_sub, sub + 2 n_sub, ... of the call's tile list (n_sub = 1: all of them).
template <int MODE>
int render_tiles(hipexec *h, const hipexec_render_cfg *c, float4 *d_rgba, int32_t *d_ids, int sub, int n_sub)
{
    constexpr int SPP = (MODE == HIPEXEC_MODE_CL) ? 1 : 4;
    const DScene &S = h->S;
    int n_all = std::min(n_tiles_of(c), h->win_hi);
    const int first = h->win_lo + sub;
    int n_tiles = n_all > first ? (n_all - first + n_sub - 1) / n_sub : 0;
    const int cx0 = std::max(0, h->rect[0]), cy0 = std::max(0, h->rect[1]);
    const int cx1 = std::min(c->width, h->rect[2]), cy1 = std::min(c->height, h->rect[3]);
    if (cx0 >= cx1 || cy0 >= cy1) return HIPEXEC_OK;
    const int x_blk0 = cx0 / 8;
    int nbx = (cx1 + 7) / 8 - x_blk0;
    int depth = (MODE == HIPEXEC_MODE_CL) ? 0 : c->max_depth;
    // tiles per batch, bounded by the shadow queue
    uint64_t samples_per_tile = (uint64_t)nbx * 64 * SPP;
    uint64_t per_tile_entries = samples_per_tile * std::max(1u, S.n_lamp);
    int tiles_per_batch = (int)std::max<uint64_t>(1, SHADOW_QUEUE_MAX_ENTRIES / per_tile_entries);
    if (h->use_bvh) tiles_per_batch = (int)std::max<uint64_t>(1, BATCH_MAX_SAMPLES / samples_per_tile);   // no shadow queue
    for (int t0 = 0; t0 < n_tiles; t0 += tiles_per_batch) {
        Frame F;
        F.W = c->width; F.H = c->height; F.tile_step = c->tile_step * n_sub;
        F.out_first = first + t0 * n_sub; F.out_step = n_sub;
        F.tile_first = c->tile_first + F.out_first * c->tile_step;
        F.n_tiles = std::min(tiles_per_batch, n_tiles - t0);
        F.nbx = nbx; F.n_slots = (uint32_t)F.n_tiles * (uint32_t)nbx * 64u; F.spp = SPP;
        F.x_blk0 = x_blk0; F.out_w = cx1 - cx0; F.cx0 = cx0; F.cx1 = cx1; F.cy0 = cy0; F.cy1 = cy1;
        uint64_t n0 = (uint64_t)F.n_slots * SPP;
        if (n0 > 0xfffffff0ull) return fail(HIPEXEC_E_LIMIT, "batch of %llu samples exceeds 32-bit indexing", (unsigned long long)n0);
        CHK(reserve_trace_buffers(h, n0, depth + 1));
        if (h->use_bvh && h->packet && h->fuse_raygen) {
            // the packet kernel makes, stores and walks the camera rays in one launch
            LAUNCH(HIPEXEC_K_PRIMARY, (k_bvh_closest_pk<MODE, true>), blocks_for(n0), S, h->B, h->rayA.as<float4>(), (uint32_t)n0,
                   h->hit0.as<float4>(), h->d_cnt64, 0, F);
        } else if (h->use_bvh) {
            LAUNCH(HIPEXEC_K_PRIMARY, (k_raygen<MODE>), blocks_for(F.n_slots), S, F, h->rayA.as<float4>(), h->hit0.as<float4>());
            CHK(launch_bvh_closest<MODE>(h, HIPEXEC_K_PRIMARY, h->rayA.as<float4>(), (uint32_t)n0, h->hit0.as<float4>(), 0));
        } else {
            LAUNCH(HIPEXEC_K_PRIMARY, (k_primary<MODE>), blocks_for(F.n_slots), S, F, h->rayA.as<float4>(), h->hit0.as<float4>(), h->d_cnt64);
        }
        int rows = 0;
        for (int k = 0; k < F.n_tiles; k++) {
            const int ty = (F.tile_first + k * F.tile_step) * HIPEXEC_TILE_ROWS;
            rows += std::max(0, std::min(ty + HIPEXEC_TILE_ROWS, cy1) - std::max(ty, cy0));
        }
        uint64_t prim = (uint64_t)rows * (uint64_t)(cx1 - cx0) * SPP;
        h->acc.rays_primary += prim;
        h->acc.tests_tri_primary += prim * S.n_tri;
        h->acc.tests_tri_closest += prim * S.n_tri;
        h->acc.tests_sph_closest += prim * S.n_sph;
        int levels_used = 1;
        const std::array<int64_t, 24> key = {1, c->width, c->height, MODE, depth, c->tile_first, c->tile_step, h->win_lo, h->win_hi,
                                             cx0, cy0, cx1, cy1, sub, n_sub, t0, (int64_t)n0, h->use_bvh, h->tail_max, h->overlap_levels,
                                             h->bvh_refill, h->fuse_shade, (int64_t)h->simple_from, (int64_t)(h->S.refract | (h->S.ao << 1))};
        CHK(trace_levels<MODE>(h, (uint32_t)n0, (size_t)n0, depth, &levels_used, key));
        LAUNCH(HIPEXEC_K_SHADE, (k_resolve<MODE>), blocks_for(F.n_slots), F, h->local.as<float4>(), (size_t)n0, levels_used,
               h->hit0.as<float4>(), c->tone, d_rgba, d_ids);
    }
    return HIPEXEC_OK;
}

int render_sub(hipexec *h, const hipexec_render_cfg *c, float4 *d_rgba, int32_t *d_ids, int sub, int n_sub)
{
    if (c->mode == HIPEXEC_MODE_CL) return render_tiles<HIPEXEC_MODE_CL>(h, c, d_rgba, d_ids, sub, n_sub);
    return render_tiles<HIPEXEC_MODE_CPU>(h, c, d_rgba, d_ids, sub, n_sub);
}

int call_begin(hipexec *h, uint32_t timing, uint32_t flags);
int create_lane_resources(hipexec *h, uint32_t n_lamp);

// Makes the lanes 1 .. want-1 of this executor if they do not exist yet (own streams, buffers, counters and host
// thread each; the scene stays owned by lane 0).
int more_lanes(hipexec *h, int want)
{
    while (h->bvh_ok && (int)h->helpers.size() + 1 < want) {
        hipexec *g = new (std::nothrow) hipexec();
        if (!g) return fail(HIPEXEC_E_NOMEM, "out of host memory");
        h->helpers.push_back(g);
        g->is_helper = true; g->device = h->device;
        g->worker = new (std::nothrow) LaneWorker();
        if (!g->worker) return fail(HIPEXEC_E_NOMEM, "out of host memory");
        g->S = h->S; g->B = h->B; g->bvh_ok = true; g->lamp_words = h->lamp_words; g->pred_home = h->pred_home;
        CHK(create_lane_resources(g, h->S.n_lamp));
        HIPCHK(hipEventCreateWithFlags(&g->ev_join, hipEventDisableTiming));
    }
    return HIPEXEC_OK;
}

int render_dispatch(hipexec *h, const hipexec_render_cfg *c, float4 *d_rgba, int32_t *d_ids)
{
    h->lanes_used = 1;
    // Several lanes pay when bounce levels exist (their small, long-tailed launches overlap the other lanes'
    // large ones); a depth-0 frame is two large uniform kernels and gains nothing.  Lane k of L traces the
    // tiles k, k + L, ... of the call's tile list; at least 4 tiles per lane.
    const bool bounces = c->mode == HIPEXEC_MODE_CPU && c->max_depth > 0;
    const int n_window = std::min(n_tiles_of(c), h->win_hi) - h->win_lo;
    int L = std::min(h->max_lanes, n_window / 4);
    if (!h->use_bvh || !bounces || L < 2) return render_sub(h, c, d_rgba, d_ids, 0, 1);
    CHK(more_lanes(h, L));
    L = std::min(L, (int)h->helpers.size() + 1);
    h->lanes_used = L;
    // fork: the other lanes' streams start after everything queued on this executor's stream so far
    HIPCHK(hipEventRecord(h->ev_fork, h->stream));
    std::vector<int> rcs((size_t)L, HIPEXEC_OK);
    std::vector<std::string> errs((size_t)L);
    for (int k = 1; k < L; k++) {
        hipexec *g = h->helpers[(size_t)k - 1];
        g->win_lo = h->win_lo; g->win_hi = h->win_hi;
        for (int q = 0; q < 4; q++) g->rect[q] = h->rect[q];
        g->size_mode = h->size_mode;
        memcpy(g->S.cam_pos, h->S.cam_pos, sizeof g->S.cam_pos); memcpy(g->S.cam_mat, h->S.cam_mat, sizeof g->S.cam_mat);
        g->block64_below = h->block64_below;
        g->refill_closest = h->refill_closest; g->sched_w = h->sched_w; g->closest_per_block = h->closest_per_block;
        g->bvh_refill = h->bvh_refill; g->shadow_waves6_from = h->shadow_waves6_from; g->bvh_default = true; g->overlap_levels = h->overlap_levels;
        g->worker->submit([&, g, k]() {
            rcs[(size_t)k] = [&]() -> int {
                HIPCHK(hipSetDevice(g->device));
                HIPCHK(hipStreamWaitEvent(g->stream, h->ev_fork, 0));
                CHK(call_begin(g, h->timing, HIPEXEC_FLAG_BVH | ((uint32_t)c->flags & (HIPEXEC_FLAG_BILINEAR | HIPEXEC_FLAG_REFRACT | HIPEXEC_FLAG_AO))));
                return render_sub(g, c, d_rgba, d_ids, k, L);
            }();
            if (rcs[(size_t)k] != HIPEXEC_OK) errs[(size_t)k] = g_err;          // g_err is per thread
        });
    }
    int rc = render_sub(h, c, d_rgba, d_ids, 0, L);
    for (int k = 1; k < L; k++) h->helpers[(size_t)k - 1]->worker->wait();
    // join: later work on this executor's stream waits for the other lanes' kernels.  Every lane is joined whatever
    // happens to one of them; the first error is the call's.
    for (int k = 1; k < L; k++) {
        hipexec *g = h->helpers[(size_t)k - 1];
        if (rcs[(size_t)k] == HIPEXEC_OK) {
            hipError_t e = hipEventRecord(g->ev_join, g->stream);
            if (e == hipSuccess) e = hipStreamWaitEvent(h->stream, g->ev_join, 0);
            if (e != hipSuccess) {
                (void)hipStreamSynchronize(g->stream);
                g->call_open = false;
                if (rc == HIPEXEC_OK) rc = fail(HIPEXEC_E_DEVICE, "joining lane %d failed: %s", k, hipGetErrorString(e));
            }
        } else {
            (void)hipStreamSynchronize(g->stream);
            g->call_open = false;
            if (rc == HIPEXEC_OK) { snprintf(g_err, sizeof g_err, "%s", errs[(size_t)k].c_str()); rc = rcs[(size_t)k]; }
        }
    }
    return rc;
}

// rays6 (host) -> ray records in rayA, sample index = position
__global__ void k_import_rays(const float *__restrict__ rays6, uint32_t n, float4 *__restrict__ recs)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *r = rays6 + 6 * (size_t)i;
    store_ray(recs, i, mk(r[0], r[1], r[2]), mk(r[3], r[4], r[5]), i, 0u);
}

// float4 RGBA pixels -> float RGB pixels (the layout of image_bitmap, image_bitmap.h:22)
__global__ void k_strip_alpha(const float4 *__restrict__ rgba, size_t n, float *__restrict__ rgb)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 c = rgba[i];
    rgb[3 * i] = c.x; rgb[3 * i + 1] = c.y; rgb[3 * i + 2] = c.z;
}

// hits[i] |= bit 0 of bits[i] (computeAnyHitsCl ORs into the caller's array, ray_tracer_cl.cpp:255,260)
__global__ void k_or_hits(const uint32_t *__restrict__ bits, uint32_t n, char *__restrict__ hits)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    hits[i] = (hits[i] || (bits[i] & 1u)) ? 1 : 0;
}

int import_rays(hipexec *h, const float *rays6, uint32_t n)
{
    CHK(h->stage_in.reserve((size_t)n * 24));
    CHK(h->rayA.reserve((size_t)n * 32));
    HIPCHK(hipMemcpyAsync(h->stage_in.p, rays6, (size_t)n * 24, hipMemcpyHostToDevice, h->stream));
    LAUNCH(HIPEXEC_K_SHADE, k_import_rays, blocks_for(n), h->stage_in.as<float>(), n, h->rayA.as<float4>());
    return HIPEXEC_OK;
}

// stream, events and counters of one lane
int create_lane_resources(hipexec *h, uint32_t n_lamp)
{
    CHK(pool_stream(h->device, &h->own_stream));
    h->stream = h->own_stream;
    HIPCHK(hipEventCreate(&h->ev_begin)); HIPCHK(hipEventCreate(&h->ev_end));
    CHK(pool_stream(h->device, &h->side));
    for (hipEvent_t *e : {&h->ev_hits[0], &h->ev_hits[1], &h->ev_lit[0], &h->ev_lit[1]})
        HIPCHK(hipEventCreateWithFlags(e, hipEventDisableTiming));
    if (const char *e = knob("HIPEXEC_PACKET")) h->packet = atoi(e) != 0;
    if (const char *e = knob("HIPEXEC_FUSE_RAYGEN")) h->fuse_raygen = atoi(e) != 0;
    if (const char *e = knob("HIPEXEC_FUSE_SHADE")) h->fuse_shade = std::max(0, std::min(2, atoi(e)));
    if (const char *e = knob("HIPEXEC_UNIFORM_FETCH")) h->uniform_fetch = (uint32_t)atoi(e) & 3u;
    if (const char *e = knob("HIPEXEC_SHADOW_WAVES6_FROM")) h->shadow_waves6_from = (uint32_t)strtoul(e, nullptr, 10);
    if (const char *e = knob("HIPEXEC_SIMPLE_FROM")) h->simple_from = (uint32_t)strtoul(e, nullptr, 10);
    if (const char *e = knob("HIPEXEC_SHADOW_DIRECT")) h->shadow_direct = atoi(e) != 0;
    if (const char *e = knob("HIPEXEC_SHADOW_PER_BLOCK")) h->shadow_per_block = (uint32_t)std::max(64, std::min(256, atoi(e) / 64 * 64));
    if (const char *e = knob("HIPEXEC_TAIL_MAX")) h->tail_max = (uint32_t)std::max(0, atoi(e));
    if (const char *e = knob("HIPEXEC_OVERLAP")) h->overlap_levels = std::max(0, std::min(2, atoi(e)));
    if (const char *e = knob("HIPEXEC_BLOCK_CLOSEST")) h->block_closest = atoi(e);
    if (const char *e = knob("HIPEXEC_BLOCK_SHADOW")) h->block_shadow = atoi(e);
    if (const char *e = knob("HIPEXEC_BLOCK_TAIL")) h->block_tail = atoi(e);
    if (const char *e = knob("HIPEXEC_BLOCK64_BELOW")) h->block64_below = (uint32_t)strtoul(e, nullptr, 10);
    h->n_cnt32 = std::max(8u, n_lamp + 1u);
    CHK(pool_alloc(h->device, false, h->n_cnt32 * sizeof(uint32_t), (void **)&h->d_cnt32));
    CHK(pool_alloc(h->device, false, CNT64_WORDS * sizeof(unsigned long long), (void **)&h->d_cnt64));
    CHK(pool_alloc(h->device, false, (size_t)LV_SLOTS * LV_PER * sizeof(uint32_t), (void **)&h->d_lvl));
    CHK(pool_alloc(h->device, true, (size_t)LV_SLOTS * LV_PER * sizeof(uint32_t), (void **)&h->h_lvl));
    if (const char *e = knob("HIPEXEC_PREDICT")) { const int v = atoi(e); h->size_default = v == 0 ? SIZE_WAIT : (v == 2 ? SIZE_BOUND : SIZE_AUTO); }
    h->size_mode = h->size_default;
    if (const char *e = knob("HIPEXEC_PREDICT_SPOIL")) h->predict_spoil = atoi(e);   // test hook, see level_end
    CHK(pool_alloc(h->device, true, h->n_cnt32 * sizeof(uint32_t), (void **)&h->h_cnt32));
    CHK(pool_alloc(h->device, true, CNT64_WORDS * sizeof(unsigned long long), (void **)&h->h_cnt64));
    HIPCHK(hipMemsetAsync(h->d_cnt32, 0, h->n_cnt32 * sizeof(uint32_t), h->stream));
    HIPCHK(hipMemsetAsync(h->d_cnt64, 0, CNT64_WORDS * sizeof(unsigned long long), h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return HIPEXEC_OK;
}

// Runs `frame` (call_begin ... call_end); if a level of one of its batches produced more rays than its launch was sized
// for (the check of call_end), the frame is traced once more with every launch sized from the bound.
template <class F>
int run_checked(hipexec *h, F frame)
{
    int rc = frame();
    if (rc == HIPEXEC_OK && h->redo) {
        h->redo = false; h->size_mode = SIZE_BOUND; h->retraced = 1u;
        rc = frame();
        h->size_mode = h->size_default; h->redo = false; h->retraced = 0u;
    }
    return rc;
}

// executor-level knobs of the environment (honoured only with HIPEXEC_KNOBS=1, see knob)
void read_executor_knobs(hipexec *h, bool *anyhit_lds_set)
{
    if (const char *e = knob("HIPEXEC_CLOSEST_R")) h->closest_r = atoi(e);
    if (const char *e = knob("HIPEXEC_ANYHIT_R")) h->anyhit_r = atoi(e);
    if (const char *e = knob("HIPEXEC_CLOSEST_LDS")) h->closest_lds = atoi(e) != 0;
    if (const char *e = knob("HIPEXEC_ANYHIT_LDS")) { h->anyhit_lds = atoi(e) != 0; *anyhit_lds_set = true; }
    if (const char *e = knob("HIPEXEC_ALGO")) h->bvh_default = strcmp(e, "scan") != 0;
    if (const char *e = knob("HIPEXEC_BVH_REFILL")) h->bvh_refill = atoi(e) != 0;
    if (const char *e = knob("HIPEXEC_REFILL_CLOSEST")) h->refill_closest = std::max(1, std::min(64, atoi(e)));
    if (const char *e = knob("HIPEXEC_CLOSEST_PER_BLOCK")) h->closest_per_block = std::max(256, std::min(16384, atoi(e)));
    if (const char *e = knob("HIPEXEC_SCHED_W")) h->sched_w = std::max(0, std::min(255, atoi(e)));
    if (const char *e = knob("HIPEXEC_LANES")) h->max_lanes = std::max(1, std::min(8, atoi(e)));      // 1: no further lanes
}

}  // namespace

// =========================================================================================
// C ABI
// =========================================================================================
extern "C" {

int hipexec_abi_version(void) { return HIPEXEC_ABI_VERSION; }

const char *hipexec_last_error(void) { return g_err; }

int hipexec_create(hipexec_t **out, const hipexec_scene_desc *scene, int device)
{
    if (!out) return fail(HIPEXEC_E_INVALID, "out handle pointer is NULL");
    *out = nullptr;
    CHK(validate_scene(scene));
    int count = 0;
    HIPCHK(hipGetDeviceCount(&count));
    if (count <= 0) return fail(HIPEXEC_E_DEVICE, "no HIP device available");
    if (device < 0) HIPCHK(hipGetDevice(&device));
    if (device >= count) return fail(HIPEXEC_E_INVALID, "device %d out of range (%d devices)", device, count);
    HIPCHK(hipSetDevice(device));
    hipexec *h = new (std::nothrow) hipexec();
    if (!h) return fail(HIPEXEC_E_NOMEM, "out of host memory");
    h->device = device;
    int rc = [&]() -> int {
        StageTrace st; g_stage = &st;
        struct Unset { ~Unset() { g_stage = nullptr; } } unset;
        h->share = new (std::nothrow) SceneShare();
        if (!h->share) return fail(HIPEXEC_E_NOMEM, "out of host memory");
        h->share->device = device; h->pred_home = h->share;
        CHK(static_scratch_init(device));
        CHK(create_lane_resources(h, scene->n_lamp));
        STAGE("lane resources");
        bool anyhit_lds_set = false;
        read_executor_knobs(h, &anyhit_lds_set);
        CHK(build_scene(h, scene));
        STAGE("rest of the scene");
        if (!h->bvh_default || !h->bvh_ok) CHK(ensure_scan_records(h));       // the scans are this executor's algorithm
        // full scans stage the lamp records through LDS (measured faster); with chunk culling each
        // wave reads only the chunks it needs and the scalar path wins
        if (!anyhit_lds_set) h->anyhit_lds = h->S.cull == 0;
        HIPCHK(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
        // (further lanes are made by the first frame that uses them, see more_lanes: streams are bound to the
        // process's few hardware queues in the order they are created, and an executor that is used with one lane,
        // as one of several with frames in flight, should not take queues it never uses)
        return HIPEXEC_OK;
    }();
    if (rc != HIPEXEC_OK) { hipexec_destroy(h); return rc; }
    *out = h;
    return HIPEXEC_OK;
}

int hipexec_clone(hipexec_t **out, hipexec_t *src)
{
    if (!out) return fail(HIPEXEC_E_INVALID, "out handle pointer is NULL");
    *out = nullptr;
    if (!src || !src->share) return fail(HIPEXEC_E_INVALID, "source handle is NULL");
    HIPCHK(hipSetDevice(src->device));
    hipexec *h = new (std::nothrow) hipexec();
    if (!h) return fail(HIPEXEC_E_NOMEM, "out of host memory");
    h->device = src->device;
    src->share->refs.fetch_add(1, std::memory_order_relaxed);
    h->share = src->share; h->pred_home = src->share;
    int rc = [&]() -> int {
        CHK(create_lane_resources(h, src->S.n_lamp));
        bool anyhit_lds_set = false;
        read_executor_knobs(h, &anyhit_lds_set);
        // the scene, the hierarchy and the source's settings; the camera-dependent scan records stay the source's own
        h->S = src->S; h->B = src->B; h->bvh_ok = src->bvh_ok; h->lamp_words = src->lamp_words;
        h->S.tri_primx = nullptr; h->S.pair_cam = nullptr; h->scan_ready = false;
        h->bvh_default = src->bvh_default; h->max_lanes = src->max_lanes;
        h->anyhit_lds = anyhit_lds_set ? h->anyhit_lds : src->anyhit_lds;
        if (!h->bvh_default || !h->bvh_ok) CHK(ensure_scan_records(h));
        HIPCHK(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
        return HIPEXEC_OK;
    }();
    if (rc != HIPEXEC_OK) { hipexec_destroy(h); return rc; }
    *out = h;
    return HIPEXEC_OK;
}

int hipexec_set_camera(hipexec_t *h, const float *cam_pos, const float *cam_mat)
{
    if (!h) return fail(HIPEXEC_E_INVALID, "handle is NULL");
    if (!cam_pos || !cam_mat) return fail(HIPEXEC_E_INVALID, "cam_pos/cam_mat is NULL");
    if (h->call_open) CHK(hipexec_sync(h));                 // a frame in flight keeps the camera it was submitted with
    memcpy(h->S.cam_pos, cam_pos, sizeof h->S.cam_pos); memcpy(h->S.cam_mat, cam_mat, sizeof h->S.cam_mat);
    // the scans' camera records are made again by the next call that scans (a set-up cost of that call; the hierarchy,
    // the default algorithm, has nothing that depends on the camera)
    h->scan_ready = false;
    if (!h->bvh_default || !h->bvh_ok) { HIPCHK(hipSetDevice(h->device)); CHK(ensure_scan_records(h)); }
    return HIPEXEC_OK;
}

int hipexec_optimize(hipexec_t *h, float *build_ms)
{
    if (build_ms) *build_ms = 0.0f;
    if (!h) return fail(HIPEXEC_E_INVALID, "handle is NULL");
    if (!h->share) return fail(HIPEXEC_E_INVALID, "a helper lane owns no scene memory");
    HIPCHK(hipSetDevice(h->device));
    if (h->call_open) CHK(hipexec_sync(h));
    if (!h->bvh_ok || h->S.n_tri <= (uint32_t)BVH_LEAF_MAX) return HIPEXEC_OK;      // no hierarchy, or a single leaf
    std::lock_guard<std::mutex> lk(h->share->m);
    if (h->share->optimized) return HIPEXEC_OK;
    // the arrays every executor of the scene traverses are rewritten in place: whatever still runs on the device ends first
    HIPCHK(hipDeviceSynchronize());
    const auto t0 = std::chrono::steady_clock::now();
    hipStream_t keep = h->stream;
    h->stream = h->own_stream;
    int rc = build_bvh(h, h->S.tri_exact, nullptr, nullptr, true, true);
    hipError_t e = hipStreamSynchronize(h->own_stream);
    h->stream = keep;
    pool_free(h->device, false, h->build_tmp);
    h->build_tmp = nullptr;
    CHK(rc);
    HIPCHK(e);
    h->share->optimized = true;
    if (build_ms) *build_ms = (float)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return HIPEXEC_OK;
}

void hipexec_destroy(hipexec_t *h)
{
    if (!h) return;
    if (h->worker) { h->worker->stop(); delete h->worker; h->worker = nullptr; }
    (void)hipSetDevice(h->device);
    for (hipexec *g : h->helpers) hipexec_destroy(g);                          // they own no scene memory
    h->helpers.clear();
    for (hipexec *g : h->prog_clones) hipexec_destroy(g);
    h->prog_clones.clear();
    pool_free(h->device, true, h->prog_pin);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->side) { (void)hipStreamSynchronize(h->side); pool_stream_back(h->device, h->side); }
    for (hipEvent_t e : {h->ev_hits[0], h->ev_hits[1], h->ev_lit[0], h->ev_lit[1]}) if (e) (void)hipEventDestroy(e);
    share_release(h->share); h->share = nullptr;                              // the last executor of the scene frees it
    pool_free(h->device, false, h->build_tmp);
    for (Buf *b : {&h->cam_primx, &h->cam_pairs, &h->rayA, &h->rayB, &h->hit0, &h->hitN, &h->hitM, &h->local, &h->occl, &h->shadowq, &h->out_rgba, &h->out_ids,
                   &h->stage_in, &h->stage_out}) b->release();
    pool_free(h->device, false, h->d_cnt32);
    pool_free(h->device, false, h->d_cnt64);
    pool_free(h->device, false, h->d_lvl);
    pool_free(h->device, true, h->h_lvl);
    pool_free(h->device, true, h->h_cnt32);
    pool_free(h->device, true, h->h_cnt64);
    for (TimedLaunch &t : h->timed) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
    for (hipEvent_t e : h->event_pool) (void)hipEventDestroy(e);
    if (h->ev_begin) (void)hipEventDestroy(h->ev_begin);
    if (h->ev_end) (void)hipEventDestroy(h->ev_end);
    if (h->own_stream) { (void)hipStreamSynchronize(h->own_stream); pool_stream_back(h->device, h->own_stream); }
    delete h;
}

int hipexec_trim(int device)
{
    int count = 0;
    HIPCHK(hipGetDeviceCount(&count));
    if (device >= count) return fail(HIPEXEC_E_INVALID, "device %d out of range (%d devices)", device, count);
    int cur = 0;
    HIPCHK(hipGetDevice(&cur));
    pool_trim(device);
    HIPCHK(hipSetDevice(cur));
    return HIPEXEC_OK;
}

int hipexec_set_stream(hipexec_t *h, void *stream)
{
    if (!h) return fail(HIPEXEC_E_INVALID, "handle is NULL");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->stream = stream ? (hipStream_t)stream : h->own_stream;
    return HIPEXEC_OK;
}

int hipexec_set_lanes(hipexec_t *h, int lanes)
{
    if (!h) return fail(HIPEXEC_E_INVALID, "handle is NULL");
    if (lanes < 1 || lanes > 8) return fail(HIPEXEC_E_INVALID, "lanes %d outside [1,8]", lanes);
    h->max_lanes = lanes;
    return HIPEXEC_OK;
}

int hipexec_set_algorithm(hipexec_t *h, int algorithm)
{
    if (!h) return fail(HIPEXEC_E_INVALID, "handle is NULL");
    if (algorithm != HIPEXEC_ALGO_SCAN && algorithm != HIPEXEC_ALGO_BVH) return fail(HIPEXEC_E_INVALID, "unknown algorithm %d", algorithm);
    h->bvh_default = algorithm == HIPEXEC_ALGO_BVH;
    if (!h->bvh_default) {                    // a set-up call may allocate; hipexec_any_hits_static later may not
        HIPCHK(hipSetDevice(h->device));
        CHK(ensure_scan_records(h));
    }
    return HIPEXEC_OK;
}

int hipexec_sync(hipexec_t *h)
{
    if (!h) return fail(HIPEXEC_E_INVALID, "handle is NULL");
    HIPCHK(hipSetDevice(h->device));
    if (h->call_open) {
        hipexec_stats *dst = h->pending_stats;
        h->pending_stats = nullptr;
        bool first = true;
        return run_checked(h, [&]() -> int {
            if (!first) {          // repeat of the asynchronous frame (hipexec_render_device) with its saved arguments
                CHK(call_begin(h, timing_mask(h->pending_cfg.flags), h->pending_cfg.flags));
                int rc = render_dispatch(h, &h->pending_cfg, h->pending_rgba, h->pending_ids);
                if (rc != HIPEXEC_OK) { h->call_open = false; return rc; }
            }
            first = false;
            return call_end(h, dst);
        });
    }
    HIPCHK(hipStreamSynchronize(h->stream));
    return HIPEXEC_OK;
}

int hipexec_render_device(hipexec_t *h, const hipexec_render_cfg *cfg, float *d_out_rgba, int32_t *d_out_ids,
                          hipexec_stats *stats)
{
    if (!h) return fail(HIPEXEC_E_INVALID, "handle is NULL");
    CHK(validate_cfg(cfg));
    if (!d_out_rgba) return fail(HIPEXEC_E_INVALID, "d_out_rgba is NULL");
    CHK(call_begin(h, timing_mask(cfg->flags), cfg->flags));
    int rc = render_dispatch(h, cfg, reinterpret_cast<float4 *>(d_out_rgba), d_out_ids);
    if (rc != HIPEXEC_OK) { h->call_open = false; return rc; }
    HIPCHK(hipEventRecord(h->ev_end, h->stream));
    h->pending_stats = stats;
    h->pending_cfg = *cfg; h->pending_rgba = reinterpret_cast<float4 *>(d_out_rgba); h->pending_ids = d_out_ids;
    return HIPEXEC_OK;
}

int hipexec_render(hipexec_t *h, const hipexec_render_cfg *cfg, float *out_rgb, int32_t *out_ids, hipexec_stats *stats)
{
    if (!h) return fail(HIPEXEC_E_INVALID, "handle is NULL");
    CHK(validate_cfg(cfg));
    if (!out_rgb) return fail(HIPEXEC_E_INVALID, "out_rgb is NULL");
    const int spp = cfg->mode == HIPEXEC_MODE_CL ? 1 : 4;
    const int n_tiles = n_tiles_of(cfg);
    const size_t rows = (size_t)n_tiles * HIPEXEC_TILE_ROWS, W = (size_t)cfg->width;
    if (n_tiles == 0) { if (stats) memset(stats, 0, sizeof *stats); return HIPEXEC_OK; }
    HIPCHK(hipSetDevice(h->device));
    CHK(h->out_rgba.reserve(rows * W * 16));
    if (out_ids) CHK(h->out_ids.reserve(rows * W * 4 * spp));
    const bool whole = cfg->tile_first == 0 && cfg->tile_step == 1;
    const size_t n_px = (size_t)cfg->height * W;
    if (whole) CHK(h->stage_out.reserve(n_px * 12));
    CHK(run_checked(h, [&]() -> int {
        CHK(call_begin(h, timing_mask(cfg->flags), cfg->flags));
        int rc = render_dispatch(h, cfg, h->out_rgba.as<float4>(), out_ids ? h->out_ids.as<int32_t>() : nullptr);
        if (rc != HIPEXEC_OK) { h->call_open = false; return rc; }
        // whole frame: rows are already in frame order; drop the alpha lane on the device
        // (24 instead of 32 bytes per pixel over PCIe, no host pass)
        if (whole) LAUNCH(HIPEXEC_K_SHADE, k_strip_alpha, blocks_for(n_px), h->out_rgba.as<float4>(), n_px, h->stage_out.as<float>());
        return call_end(h, stats);
    }));
    if (whole) {
        // copy straight into the caller's image
        HIPCHK(hipMemcpy(out_rgb, h->stage_out.p, n_px * 12, hipMemcpyDeviceToHost));
        if (out_ids) HIPCHK(hipMemcpy(out_ids, h->out_ids.p, n_px * spp * sizeof(int32_t), hipMemcpyDeviceToHost));
        return HIPEXEC_OK;
    }
    std::vector<float> rgba(rows * W * 4);
    HIPCHK(hipMemcpy(rgba.data(), h->out_rgba.p, rgba.size() * 4, hipMemcpyDeviceToHost));
    std::vector<int32_t> ids;
    if (out_ids) { ids.resize(rows * W * spp); HIPCHK(hipMemcpy(ids.data(), h->out_ids.p, ids.size() * 4, hipMemcpyDeviceToHost)); }
    for (int k = 0; k < n_tiles; k++) {
        for (int ly = 0; ly < HIPEXEC_TILE_ROWS; ly++) {
            int y = (cfg->tile_first + k * cfg->tile_step) * HIPEXEC_TILE_ROWS + ly;
            if (y >= cfg->height) break;
            size_t lrow = (size_t)k * HIPEXEC_TILE_ROWS + ly;
            for (size_t x = 0; x < W; x++) {
                const float *src = &rgba[(lrow * W + x) * 4];
                float *dst = out_rgb + ((size_t)y * W + x) * 3;
                dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
            }
            if (out_ids) memcpy(out_ids + (size_t)y * W * spp, &ids[lrow * W * spp], W * spp * sizeof(int32_t));
        }
    }
    return HIPEXEC_OK;
}

// One step of a progressive frame on executor `e`: the tiles [t0, t1) traced, their alpha stripped and the rows copied
// into e's pinned staging, all queued on e's stream (nothing waits here).
static int prog_submit(hipexec *e, const hipexec_render_cfg *cfg, int t0, int t1)
{
    const size_t W = (size_t)cfg->width;
    const int y0 = t0 * HIPEXEC_TILE_ROWS, rows = std::min(cfg->height, t1 * HIPEXEC_TILE_ROWS) - y0;
    const size_t n_px = (size_t)rows * W;
    hipexec *h = e;                                            // LAUNCH works on `h`
    h->win_lo = t0; h->win_hi = t1;
    CHK(call_begin(h, timing_mask(cfg->flags), cfg->flags));
    int r = render_dispatch(h, cfg, h->out_rgba.as<float4>(), nullptr);
    if (r != HIPEXEC_OK) { h->call_open = false; return r; }
    LAUNCH(HIPEXEC_K_SHADE, k_strip_alpha, blocks_for(n_px), h->out_rgba.as<float4>() + (size_t)y0 * W, n_px, h->stage_out.as<float>());
    HIPCHK(hipMemcpyAsync(h->prog_pin, h->stage_out.p, n_px * 12, hipMemcpyDeviceToHost, h->stream));
    return HIPEXEC_OK;
}

int hipexec_render_progressive(hipexec_t *h, const hipexec_render_cfg *cfg, int32_t rows_per_step, float *out_rgb,
                               hipexec_rows_cb on_rows, void *user, hipexec_stats *stats)
{
    if (!h) return fail(HIPEXEC_E_INVALID, "handle is NULL");
    CHK(validate_cfg(cfg));
    if (!out_rgb) return fail(HIPEXEC_E_INVALID, "out_rgb is NULL");
    if (cfg->tile_first != 0 || cfg->tile_step != 1) return fail(HIPEXEC_E_INVALID, "progressive rendering takes whole frames (tile_first 0, tile_step 1)");
    if (rows_per_step < 1) return fail(HIPEXEC_E_INVALID, "rows_per_step %d is not positive", rows_per_step);
    if (h->call_open) CHK(hipexec_sync(h));
    const int n_tiles = n_tiles_of(cfg);
    const int tiles_per_step = std::max(1, (rows_per_step + HIPEXEC_TILE_ROWS - 1) / HIPEXEC_TILE_ROWS);
    const int n_steps = (n_tiles + tiles_per_step - 1) / tiles_per_step;
    const size_t W = (size_t)cfg->width;
    HIPCHK(hipSetDevice(h->device));
    // Steps are kept in flight on up to four executors (HIPEXEC_PROG_IN_FLIGHT) of this scene (this one and clones of it, which share
    // the scene's device memory and are kept for later calls), one lane each: a step is a small frame whose launches have
    // floors of their own (the longest ray of each level), and while the rows of one step are copied and handed to the
    // caller the next steps are already being traced.  Steps are delivered in order.
    // (measured on C2, 1080p, whole frame in host memory, hipexec_render = 5.2 ms: steps of 48 rows 13.1 ms one at a time,
    // 7.6 / 8.9 / 7.4 / 7.6 ms with 2 / 3 / 4 / 6 in flight and the process's default of four hardware queues, 5.8 ms with
    // 6 in flight and GPU_MAX_HW_QUEUES=8; steps of 136 rows 7.5 -> 5.0 ms, of 272 rows 6.1 -> 4.9 ms)
    int in_flight = 4;
    if (const char *e = knob("HIPEXEC_PROG_IN_FLIGHT")) in_flight = std::max(1, std::min(8, atoi(e)));
    const int K = std::max(1, std::min(in_flight, n_steps));
    while ((int)h->prog_clones.size() + 1 < K) {
        hipexec *c = nullptr;
        CHK(hipexec_clone(&c, h));
        h->prog_clones.push_back(c);
    }
    std::vector<hipexec *> ex((size_t)K);
    ex[0] = h;
    for (int k = 1; k < K; k++) ex[(size_t)k] = h->prog_clones[(size_t)k - 1];
    const size_t step_bytes = (size_t)tiles_per_step * HIPEXEC_TILE_ROWS * W * 12;
    struct RestoreLanes { hipexec *h; int lanes; ~RestoreLanes() { h->max_lanes = lanes; } } restore_lanes{h, h->max_lanes};
    for (hipexec *e : ex) {
        // the clones follow this executor's camera, algorithm and knobs of the moment
        if (e != h) {
            if (memcmp(e->S.cam_pos, h->S.cam_pos, sizeof e->S.cam_pos) || memcmp(e->S.cam_mat, h->S.cam_mat, sizeof e->S.cam_mat)) {
                memcpy(e->S.cam_pos, h->S.cam_pos, sizeof e->S.cam_pos); memcpy(e->S.cam_mat, h->S.cam_mat, sizeof e->S.cam_mat);
                e->scan_ready = false;                         // its scan records belong to the old camera
            }
            e->bvh_default = h->bvh_default; e->size_default = h->size_default; e->size_mode = h->size_default;
        }
        if (K > 1) e->max_lanes = 1;
        CHK(e->out_rgba.reserve((size_t)n_tiles * HIPEXEC_TILE_ROWS * W * 16));
        CHK(e->stage_out.reserve(step_bytes));
        if (e->prog_pin_cap < step_bytes) {
            pool_free(e->device, true, e->prog_pin); e->prog_pin = nullptr; e->prog_pin_cap = 0;
            CHK(pool_alloc(e->device, true, step_bytes, &e->prog_pin));
            e->prog_pin_cap = step_bytes;
        }
    }
    hipexec_stats total{};
    int rc = HIPEXEC_OK;
    // collects step `k` (waits for its executor's stream: trace and copy), repeats it if a launch was sized too small,
    // hands its rows over
    auto deliver = [&](int k) -> int {
        hipexec *e = ex[(size_t)(k % K)];
        const int t0 = k * tiles_per_step, t1 = std::min(n_tiles, t0 + tiles_per_step);
        const int y0 = t0 * HIPEXEC_TILE_ROWS, rows = std::min(cfg->height, t1 * HIPEXEC_TILE_ROWS) - y0;
        hipexec_stats st{};
        bool first = true;
        CHK(run_checked(e, [&]() -> int {
            if (!first) CHK(prog_submit(e, cfg, t0, t1));
            first = false;
            return call_end(e, &st);
        }));
        memcpy(out_rgb + (size_t)y0 * W * 3, e->prog_pin, (size_t)rows * W * 12);
        add_stats(total, st);
        total.total_ms += st.total_ms; total.algorithm = st.algorithm; total.lanes = std::max(total.lanes, st.lanes);
        if (on_rows) on_rows(user, y0, rows);
        return HIPEXEC_OK;
    };
    int submitted = 0, delivered = 0;
    for (; submitted < n_steps && rc == HIPEXEC_OK; submitted++) {
        if (submitted - delivered >= K) { rc = deliver(delivered); delivered++; if (rc != HIPEXEC_OK) break; }
        const int t0 = submitted * tiles_per_step;
        rc = prog_submit(ex[(size_t)(submitted % K)], cfg, t0, std::min(n_tiles, t0 + tiles_per_step));
    }
    for (; delivered < submitted && rc == HIPEXEC_OK; delivered++) rc = deliver(delivered);
    for (hipexec *e : ex) {
        if (e->call_open) { (void)hipStreamSynchronize(e->stream); e->call_open = false; }      // after an error: nothing stays in flight
        e->win_lo = 0; e->win_hi = INT_MAX;
    }
    if (rc == HIPEXEC_OK && stats) *stats = total;
    return rc;
}

int hipexec_render_rect(hipexec_t *h, const hipexec_render_cfg *cfg, int32_t x, int32_t y, int32_t w, int32_t hgt,
                        float *out_rgb, int32_t *out_ids, hipexec_stats *stats)
{
    if (!h) return fail(HIPEXEC_E_INVALID, "handle is NULL");
    CHK(validate_cfg(cfg));
    if (!out_rgb) return fail(HIPEXEC_E_INVALID, "out_rgb is NULL");
    if (cfg->tile_first != 0 || cfg->tile_step != 1) return fail(HIPEXEC_E_INVALID, "a rectangle is a window of the whole frame (tile_first 0, tile_step 1)");
    if (w < 0 || hgt < 0) return fail(HIPEXEC_E_INVALID, "rectangle size %dx%d is negative", w, hgt);
    if (stats) memset(stats, 0, sizeof *stats);
    // pixels outside the frame are dropped, like image_bitmap::setPixel does (image_bitmap.h:36-49)
    const int x0 = std::max(0, x), y0 = std::max(0, y);
    const int x1 = (int)std::min<int64_t>(cfg->width, (int64_t)x + w), y1 = (int)std::min<int64_t>(cfg->height, (int64_t)y + hgt);
    if (x0 >= x1 || y0 >= y1) return HIPEXEC_OK;
    const int spp = cfg->mode == HIPEXEC_MODE_CL ? 1 : 4;
    const int t_lo = y0 / HIPEXEC_TILE_ROWS, t_hi = (y1 + HIPEXEC_TILE_ROWS - 1) / HIPEXEC_TILE_ROWS;
    const size_t rw = (size_t)(x1 - x0), rows_all = (size_t)t_hi * HIPEXEC_TILE_ROWS;      // output rows are addressed by frame row
    HIPCHK(hipSetDevice(h->device));
    CHK(h->out_rgba.reserve(rows_all * rw * 16));
    if (out_ids) CHK(h->out_ids.reserve(rows_all * rw * 4 * spp));
    CHK(h->stage_out.reserve((size_t)(y1 - y0) * rw * 12));
    h->win_lo = t_lo; h->win_hi = t_hi;
    h->rect[0] = x0; h->rect[1] = y0; h->rect[2] = x1; h->rect[3] = y1;
    int rc = [&]() -> int {
        const size_t n_px = (size_t)(y1 - y0) * rw;
        CHK(run_checked(h, [&]() -> int {
            CHK(call_begin(h, timing_mask(cfg->flags), cfg->flags));
            int r = render_dispatch(h, cfg, h->out_rgba.as<float4>(), out_ids ? h->out_ids.as<int32_t>() : nullptr);
            if (r != HIPEXEC_OK) { h->call_open = false; return r; }
            LAUNCH(HIPEXEC_K_SHADE, k_strip_alpha, blocks_for(n_px), h->out_rgba.as<float4>() + (size_t)y0 * rw, n_px, h->stage_out.as<float>());
            return call_end(h, stats);
        }));
        HIPCHK(hipMemcpy2D(out_rgb + ((size_t)y0 * cfg->width + x0) * 3, (size_t)cfg->width * 12, h->stage_out.p, rw * 12, rw * 12,
                           (size_t)(y1 - y0), hipMemcpyDeviceToHost));
        if (out_ids)
            HIPCHK(hipMemcpy2D(out_ids + ((size_t)y0 * cfg->width + x0) * spp, (size_t)cfg->width * 4 * spp,
                               h->out_ids.as<int32_t>() + (size_t)y0 * rw * spp, rw * 4 * spp, rw * 4 * spp, (size_t)(y1 - y0),
                               hipMemcpyDeviceToHost));
        return HIPEXEC_OK;
    }();
    h->win_lo = 0; h->win_hi = INT_MAX;
    h->rect[0] = 0; h->rect[1] = 0; h->rect[2] = INT_MAX; h->rect[3] = INT_MAX;
    return rc;
}

int hipexec_closest_hits(hipexec_t *h, int mode, const float *rays6, uint32_t n, hipexec_hit *out)
{
    if (!h) return fail(HIPEXEC_E_INVALID, "handle is NULL");
    if (mode != HIPEXEC_MODE_CPU && mode != HIPEXEC_MODE_CL) return fail(HIPEXEC_E_INVALID, "unknown mode %d", mode);
    if (n == 0) return HIPEXEC_OK;
    if (!rays6 || !out) return fail(HIPEXEC_E_INVALID, "rays6/out is NULL");
    CHK(call_begin(h, false, 0));
    for (uint32_t off = 0; off < n; off += BATCH_API_CHUNK) {
        uint32_t m = std::min(BATCH_API_CHUNK, n - off);
        CHK(import_rays(h, rays6 + 6 * (size_t)off, m));
        CHK(h->hit0.reserve((size_t)m * 16));
        CHK(h->stage_out.reserve((size_t)m * sizeof(hipexec_hit)));
        if (mode == HIPEXEC_MODE_CL) CHK(launch_closest<HIPEXEC_MODE_CL>(h, h->rayA.as<float4>(), m, h->hit0.as<float4>()));
        else CHK(launch_closest<HIPEXEC_MODE_CPU>(h, h->rayA.as<float4>(), m, h->hit0.as<float4>()));
        LAUNCH(HIPEXEC_K_SHADE, k_export_hits, blocks_for(m), h->S, h->rayA.as<float4>(), h->hit0.as<float4>(), m,
               h->stage_out.as<hipexec_hit>());
        HIPCHK(hipMemcpyAsync(out + off, h->stage_out.p, (size_t)m * sizeof(hipexec_hit), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return call_end(h, nullptr);
}

int hipexec_any_hits(hipexec_t *h, int mode, const float *rays6, uint32_t n, char *hits)
{
    if (!h) return fail(HIPEXEC_E_INVALID, "handle is NULL");
    if (mode != HIPEXEC_MODE_CPU && mode != HIPEXEC_MODE_CL) return fail(HIPEXEC_E_INVALID, "unknown mode %d", mode);
    if (n == 0) return HIPEXEC_OK;
    if (!rays6 || !hits) return fail(HIPEXEC_E_INVALID, "rays6/hits is NULL");
    CHK(call_begin(h, false, 0));
    std::vector<uint32_t> bits;
    for (uint32_t off = 0; off < n; off += BATCH_API_CHUNK) {
        uint32_t m = std::min(BATCH_API_CHUNK, n - off);
        CHK(import_rays(h, rays6 + 6 * (size_t)off, m));       // sample index = position, lamp = 0
        CHK(h->stage_out.reserve((size_t)m * 4));
        HIPCHK(hipMemsetAsync(h->stage_out.p, 0, (size_t)m * 4, h->stream));
        if (mode == HIPEXEC_MODE_CL) CHK(launch_anyhit<HIPEXEC_MODE_CL>(h, h->rayA.as<float4>(), m, h->stage_out.as<uint32_t>(), 1u));
        else CHK(launch_anyhit<HIPEXEC_MODE_CPU>(h, h->rayA.as<float4>(), m, h->stage_out.as<uint32_t>(), 1u));
        bits.resize(m);
        HIPCHK(hipMemcpyAsync(bits.data(), h->stage_out.p, (size_t)m * 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        for (uint32_t i = 0; i < m; i++) hits[off + i] = (hits[off + i] || (bits[i] & 1u)) ? 1 : 0;
    }
    return call_end(h, nullptr);
}

int hipexec_any_hits_static(hipexec_t *h, int mode, const float *d_rays6, uint32_t n, char *d_hits)
{
    if (!h) return fail(HIPEXEC_E_INVALID, "handle is NULL");
    if (mode != HIPEXEC_MODE_CPU && mode != HIPEXEC_MODE_CL) return fail(HIPEXEC_E_INVALID, "unknown mode %d", mode);
    if (n == 0) return HIPEXEC_OK;
    if (!d_rays6 || !d_hits) return fail(HIPEXEC_E_INVALID, "d_rays6/d_hits is NULL");
    if (n > HIPEXEC_STATIC_MAX_RAYS)
        return fail(HIPEXEC_E_LIMIT, "%u rays exceed the %u of the fixed work memory (HIPEXEC_STATIC_MAX_RAYS)", n, HIPEXEC_STATIC_MAX_RAYS);
    unsigned char *base = (h->device >= 0 && h->device < STATIC_MAX_DEVICES) ? g_static_ptr[h->device] : nullptr;
    if (!base) return fail(HIPEXEC_E_DEVICE, "no fixed work memory on device %d", h->device);
    // everything that can refuse the call comes before the first enqueue: a refused call leaves the shared array alone
    const bool bvh = h->bvh_default && h->bvh_ok;
    if (!bvh && !h->scan_ready)
        return fail(HIPEXEC_E_INVALID, "the scan algorithm's records do not exist yet and this call cannot allocate them: run any other "
                                       "call of this executor with the scan algorithm first");
    HIPCHK(hipSetDevice(h->device));
    unsigned long long *cnt = reinterpret_cast<unsigned long long *>(base);
    float4 *recs = reinterpret_cast<float4 *>(base + STATIC_CNT_BYTES);
    uint32_t *bits = reinterpret_cast<uint32_t *>(base + STATIC_CNT_BYTES + STATIC_REC_BYTES);
    hipStream_t st = h->stream;
    // the array keeps the last call's contents: counters and occlusion words start at zero, the records are overwritten
    HIPCHK(hipMemsetAsync(cnt, 0, STATIC_CNT_BYTES, st));
    HIPCHK(hipMemsetAsync(bits, 0, (size_t)n * 4, st));
    const dim3 blk(256), grid(blocks_for(n));
    hipLaunchKernelGGL(k_import_rays, grid, blk, 0, st, d_rays6, n, recs);
    HIPCHK(hipGetLastError());
    if (bvh) {
        if (mode == HIPEXEC_MODE_CL) hipLaunchKernelGGL((k_bvh_anyhit<HIPEXEC_MODE_CL>), grid, blk, 0, st, h->S, h->B, recs, n, bits, 1u, cnt);
        else hipLaunchKernelGGL((k_bvh_anyhit<HIPEXEC_MODE_CPU>), grid, blk, 0, st, h->S, h->B, recs, n, bits, 1u, cnt);
    } else {
        constexpr int R = 2;
        const dim3 sgrid(blocks_for(((uint64_t)n + 64 * R - 1) / (64 * R) * 64));
        DScene S = h->S;
        S.bilinear = 0u;
        if (mode == HIPEXEC_MODE_CL) hipLaunchKernelGGL((k_anyhit<HIPEXEC_MODE_CL, R>), sgrid, blk, 0, st, S, recs, n, bits, 1u, cnt);
        else hipLaunchKernelGGL((k_anyhit<HIPEXEC_MODE_CPU, R>), sgrid, blk, 0, st, S, recs, n, bits, 1u, cnt);
    }
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(k_or_hits, grid, blk, 0, st, bits, n, d_hits);
    HIPCHK(hipGetLastError());
    return HIPEXEC_OK;
}

int hipexec_trace_rays(hipexec_t *h, int mode, int max_depth, const float *rays6, uint32_t n, float *colours3,
                       hipexec_stats *stats)
{
    if (!h) return fail(HIPEXEC_E_INVALID, "handle is NULL");
    if (mode != HIPEXEC_MODE_CPU && mode != HIPEXEC_MODE_CL) return fail(HIPEXEC_E_INVALID, "unknown mode %d", mode);
    if (max_depth < 0 || max_depth > 64) return fail(HIPEXEC_E_LIMIT, "max_depth %d outside [0,64]", max_depth);
    if (n == 0) { if (stats) memset(stats, 0, sizeof *stats); return HIPEXEC_OK; }
    if (!rays6 || !colours3) return fail(HIPEXEC_E_INVALID, "rays6/colours3 is NULL");
    CHK(call_begin(h, false, 0));
    h->size_mode = SIZE_WAIT;                     // a synchronous call on the caller's rays: the counts are read back level by level
    const std::array<int64_t, 24> no_key{};
    int rc = [&]() -> int {
    const DScene &S = h->S;
    int depth = mode == HIPEXEC_MODE_CL ? 0 : max_depth;
    uint32_t chunk = (uint32_t)std::max<uint64_t>(1024, std::min<uint64_t>(BATCH_API_CHUNK, SHADOW_QUEUE_MAX_ENTRIES / std::max(1u, S.n_lamp)));
    for (uint32_t off = 0; off < n; off += chunk) {
        uint32_t m = std::min(chunk, n - off);
        CHK(reserve_trace_buffers(h, m, depth + 1));
        CHK(import_rays(h, rays6 + 6 * (size_t)off, m));
        CHK(h->stage_out.reserve((size_t)m * 12));
        HIPCHK(hipMemcpyAsync(h->stage_out.p, colours3 + 3 * (size_t)off, (size_t)m * 12, hipMemcpyHostToDevice, h->stream));
        h->acc.rays_primary += m;
        h->acc.tests_tri_closest += (uint64_t)m * S.n_tri; h->acc.tests_tri_primary += (uint64_t)m * S.n_tri;
        h->acc.tests_sph_closest += (uint64_t)m * S.n_sph;
        int levels_used = 1;
        if (mode == HIPEXEC_MODE_CL) {
            CHK(launch_closest<HIPEXEC_MODE_CL>(h, h->rayA.as<float4>(), m, h->hit0.as<float4>()));
            CHK(trace_levels<HIPEXEC_MODE_CL>(h, m, m, depth, &levels_used, no_key));
        } else {
            CHK(launch_closest<HIPEXEC_MODE_CPU>(h, h->rayA.as<float4>(), m, h->hit0.as<float4>()));
            CHK(trace_levels<HIPEXEC_MODE_CPU>(h, m, m, depth, &levels_used, no_key));
        }
        LAUNCH(HIPEXEC_K_SHADE, k_resolve_batch, blocks_for(m), m, h->local.as<float4>(), (size_t)m, levels_used,
               h->stage_out.as<float>());
        HIPCHK(hipMemcpyAsync(colours3 + 3 * (size_t)off, h->stage_out.p, (size_t)m * 12, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return call_end(h, stats);
    }();
    h->size_mode = h->size_default;
    return rc;
}

}  // extern "C"
