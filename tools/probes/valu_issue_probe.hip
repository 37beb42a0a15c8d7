// How many vector instructions a SIMD of gfx950 issues per cycle for the instruction forms of the hierarchy walk's
// box test, with 1..8 waves per SIMD.  Every kernel runs ITER iterations of an unrolled block of 32 instructions of one
// form on its own registers (no memory access inside the loop); a wave stamps s_memtime around the loop, and the
// figure printed is waves-per-SIMD * instructions / cycles of the slowest wave.
// build: hipcc -O3 --offload-arch=gfx950 -o build_ab/valu_issue_probe tools/probes/valu_issue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define ITER 2000

#define REP8(X) X X X X X X X X
#define REP4(X) X X X X

// forms: each macro is 8 instructions on accumulators a0..a7 (independent of one another)
#define F_FMA   "v_fma_f32 %0, %8, %9, %0\n v_fma_f32 %1, %8, %9, %1\n v_fma_f32 %2, %8, %9, %2\n v_fma_f32 %3, %8, %9, %3\n" \
                "v_fma_f32 %4, %8, %9, %4\n v_fma_f32 %5, %8, %9, %5\n v_fma_f32 %6, %8, %9, %6\n v_fma_f32 %7, %8, %9, %7\n"
#define F_ADD   "v_add_f32 %0, %8, %0\n v_add_f32 %1, %8, %1\n v_add_f32 %2, %8, %2\n v_add_f32 %3, %8, %3\n" \
                "v_add_f32 %4, %8, %4\n v_add_f32 %5, %8, %5\n v_add_f32 %6, %8, %6\n v_add_f32 %7, %8, %7\n"
#define F_ADDABS_S "v_add_f32_e64 %0, |%0|, %10\n v_add_f32_e64 %1, |%1|, %10\n v_add_f32_e64 %2, |%2|, %10\n v_add_f32_e64 %3, |%3|, %10\n" \
                "v_add_f32_e64 %4, |%4|, %10\n v_add_f32_e64 %5, |%5|, %10\n v_add_f32_e64 %6, |%6|, %10\n v_add_f32_e64 %7, |%7|, %10\n"
#define F_SUB_S "v_sub_f32 %0, %10, %0\n v_sub_f32 %1, %10, %1\n v_sub_f32 %2, %10, %2\n v_sub_f32 %3, %10, %3\n" \
                "v_sub_f32 %4, %10, %4\n v_sub_f32 %5, %10, %5\n v_sub_f32 %6, %10, %6\n v_sub_f32 %7, %10, %7\n"
#define F_FMAABS "v_fma_f32 %0, -%8, |%9|, %0\n v_fma_f32 %1, -%8, |%9|, %1\n v_fma_f32 %2, -%8, |%9|, %2\n v_fma_f32 %3, -%8, |%9|, %3\n" \
                "v_fma_f32 %4, -%8, |%9|, %4\n v_fma_f32 %5, -%8, |%9|, %5\n v_fma_f32 %6, -%8, |%9|, %6\n v_fma_f32 %7, -%8, |%9|, %7\n"
#define F_MAX3  "v_max3_f32 %0, %0, %8, %9\n v_max3_f32 %1, %1, %8, %9\n v_max3_f32 %2, %2, %8, %9\n v_max3_f32 %3, %3, %8, %9\n" \
                "v_max3_f32 %4, %4, %8, %9\n v_max3_f32 %5, %5, %8, %9\n v_max3_f32 %6, %6, %8, %9\n v_max3_f32 %7, %7, %8, %9\n"
#define F_CMPV  "v_cmp_le_f32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %9, vcc\n v_cmp_le_f32 vcc, %1, %8\n v_cndmask_b32 %1, %1, %9, vcc\n" \
                "v_cmp_le_f32 vcc, %2, %8\n v_cndmask_b32 %2, %2, %9, vcc\n v_cmp_le_f32 vcc, %3, %8\n v_cndmask_b32 %3, %3, %9, vcc\n"
// three compares into scalar pairs and their conjunction, the tail of a slot test (3 vector + 2 scalar instructions), twice, + 2 adds
#define F_CMP3  "v_cmp_le_f32 vcc, %0, %8\n v_cmp_le_f32_e64 s[20:21], 0, %1\n v_cmp_le_f32_e64 s[22:23], %2, %9\n s_and_b64 vcc, vcc, s[20:21]\n s_and_b64 vcc, vcc, s[22:23]\n v_cndmask_b32 %3, %3, %9, vcc\n" \
                "v_cmp_le_f32 vcc, %4, %8\n v_cmp_le_f32_e64 s[20:21], 0, %5\n v_cmp_le_f32_e64 s[22:23], %6, %9\n s_and_b64 vcc, vcc, s[20:21]\n s_and_b64 vcc, vcc, s[22:23]\n v_cndmask_b32 %7, %7, %9, vcc\n"
// the same decision with one compare: max(tmin, 0) <= min(tmax, limit)
#define F_CMP1  "v_max_f32 %0, 0, %0\n v_min_f32 %1, %1, %8\n v_cmp_le_f32 vcc, %0, %1\n v_cndmask_b32 %3, %3, %9, vcc\n" \
                "v_max_f32 %4, 0, %4\n v_min_f32 %5, %5, %8\n v_cmp_le_f32 vcc, %4, %5\n v_cndmask_b32 %7, %7, %9, vcc\n"
// a branch over nothing after a compare: the pattern of the conditional pushes (v_cmp, s_and_saveexec, s_cbranch_execz, s_or exec)
#define F_EXEC  "v_cmp_le_f32 vcc, %0, %8\n s_and_saveexec_b64 s[20:21], vcc\n v_add_f32 %1, %8, %1\n s_or_b64 exec, exec, s[20:21]\n" \
                "v_cmp_le_f32 vcc, %2, %8\n s_and_saveexec_b64 s[20:21], vcc\n v_add_f32 %3, %8, %3\n s_or_b64 exec, exec, s[20:21]\n"

#define KERNEL(NAME, BODY, NV)                                                                                   \
__global__ __launch_bounds__(1024) void NAME(float *out, unsigned long long *cyc, float x, float y, float s)      \
{                                                                                                                \
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    float vx = x + threadIdx.x * 1e-9f, vy = y;                                                                  \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                        \
    for (int i = 0; i < ITER; i++)                                                                               \
        asm volatile(REP4(BODY) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                     : "v"(vx), "v"(vy), "s"(s) : "vcc", "scc", "s20", "s21", "s22", "s23");                            \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                        \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                 \
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t1 - t0;                               \
}

KERNEL(k_fma, F_FMA, 8)
KERNEL(k_add, F_ADD, 8)
KERNEL(k_addabs_s, F_ADDABS_S, 8)
KERNEL(k_sub_s, F_SUB_S, 8)
KERNEL(k_fmaabs, F_FMAABS, 8)
KERNEL(k_max3, F_MAX3, 8)
KERNEL(k_cmpv, F_CMPV, 8)
KERNEL(k_cmp3, F_CMP3, 8)
KERNEL(k_cmp1, F_CMP1, 8)
KERNEL(k_exec, F_EXEC, 6)


// packed FP32 (two lanes' worth of work per instruction: the scan kernels' filter): accumulators are register pairs
typedef float v2f __attribute__((ext_vector_type(2)));
#define F_PKFMA "v_pk_fma_f32 %0, %8, %9, %0\n v_pk_fma_f32 %1, %8, %9, %1\n v_pk_fma_f32 %2, %8, %9, %2\n v_pk_fma_f32 %3, %8, %9, %3\n" \
                "v_pk_fma_f32 %4, %8, %9, %4\n v_pk_fma_f32 %5, %8, %9, %5\n v_pk_fma_f32 %6, %8, %9, %6\n v_pk_fma_f32 %7, %8, %9, %7\n"
#define F_PKMUL "v_pk_mul_f32 %0, %8, %0\n v_pk_mul_f32 %1, %8, %1\n v_pk_mul_f32 %2, %8, %2\n v_pk_mul_f32 %3, %8, %3\n" \
                "v_pk_mul_f32 %4, %8, %4\n v_pk_mul_f32 %5, %8, %5\n v_pk_mul_f32 %6, %8, %6\n v_pk_mul_f32 %7, %8, %7\n"
#define KERNEL2(NAME, BODY)                                                                                      \
__global__ __launch_bounds__(1024) void NAME(float *out, unsigned long long *cyc, float x, float y, float s)     \
{                                                                                                                \
    v2f a0 = {(float)threadIdx.x, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f; \
    v2f vx = {x + threadIdx.x * 1e-9f, x}, vy = {y, y};                                                          \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                        \
    for (int i = 0; i < ITER; i++)                                                                               \
        asm volatile(REP4(BODY) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                     : "v"(vx), "v"(vy), "s"(s) : "vcc", "scc");                                                 \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                        \
    v2f t = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                                               \
    out[blockIdx.x * blockDim.x + threadIdx.x] = t.x + t.y;                                                      \
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t1 - t0;                    \
}
KERNEL2(k_pkfma, F_PKFMA)
KERNEL2(k_pkmul, F_PKMUL)

typedef void (*kern_t)(float *, unsigned long long *, float, float, float);

int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    struct Form { const char *name; kern_t k; int valu; int salu; } forms[] = {
        {"v_fma_f32 v,v,v", k_fma, 8, 0}, {"v_add_f32 v,v", k_add, 8, 0}, {"v_add_f32 |v|, s (VOP3, abs, SGPR)", k_addabs_s, 8, 0},
        {"v_sub_f32 s, v (VOP2, SGPR)", k_sub_s, 8, 0}, {"v_fma_f32 -v,|v|,v", k_fmaabs, 8, 0}, {"v_max3_f32", k_max3, 8, 0},
        {"v_cmp vcc + v_cndmask", k_cmpv, 8, 0}, {"3 v_cmp + 2 s_and + v_cndmask", k_cmp3, 8, 4}, {"v_max, v_min, v_cmp, v_cndmask", k_cmp1, 8, 0},
        {"v_cmp, s_and_saveexec, v_add, s_or exec", k_exec, 4, 4},
        {"v_pk_fma_f32 (counted once)", k_pkfma, 8, 0}, {"v_pk_mul_f32 (counted once)", k_pkmul, 8, 0}};
    // w waves per SIMD on every CU: blocks of 256 * w threads (a block's waves are spread over the CU's four SIMDs) with
    // enough dynamic LDS that one block (w <= 4) or two (w = 6, 8) fit a CU, one or two blocks per CU in the grid
    struct Cfg { int w, threads, lds, blocks; } cfgs[] = {{1, 256, 100 << 10, 256}, {2, 512, 100 << 10, 256}, {3, 768, 100 << 10, 256},
                                                         {4, 1024, 100 << 10, 256}, {6, 768, 70 << 10, 512}, {8, 1024, 70 << 10, 512}};
    float *out; unsigned long long *cyc;
    if (hipMalloc(&out, (size_t)512 * 1024 * 4) != hipSuccess || hipMalloc(&cyc, (size_t)512 * 16 * 8) != hipSuccess) return 1;
    std::vector<unsigned long long> h((size_t)512 * 16);
    hipEvent_t ev0, ev1; hipEventCreate(&ev0); hipEventCreate(&ev1);
    double rate[6] = {0};
    printf("%-42s", "vector instructions per cycle and SIMD");
    for (const Cfg &c : cfgs) printf("  %d waves", c.w);
    printf("\n");
    for (const Form &f : forms) {
        if (hipFuncSetAttribute((const void *)f.k, hipFuncAttributeMaxDynamicSharedMemorySize, 100 << 10) != hipSuccess) return 2;
        printf("%-42s", f.name);
        for (const Cfg &c : cfgs) {
            float ms = 0.f;
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(ev0, 0);
                hipLaunchKernelGGL(f.k, dim3(c.blocks), dim3(c.threads), c.lds, 0, out, cyc, 1.0001f, 0.9999f, 0.5f);
                hipEventRecord(ev1, 0);
                if (hipDeviceSynchronize() != hipSuccess) return 3;
                hipEventElapsedTime(&ms, ev0, ev1);
            }
            const size_t waves = (size_t)c.blocks * c.threads / 64;
            if (hipMemcpy(h.data(), cyc, waves * 8, hipMemcpyDeviceToHost) != hipSuccess) return 4;
            std::sort(h.begin(), h.begin() + (long)waves);
            printf("  %7.3f", (double)c.w * f.valu * 4 * ITER / (double)h[waves / 2]);
            // s_memtime ticks of the slowest wave per second of the launch's event time: the counter's rate if the launch is one round of waves
            if (&f == &forms[0]) rate[&c - cfgs] = (double)h[waves - 1] / (ms * 1e-3) / 1e9;
        }
        printf("\n");
    }
    printf("%-42s", "ticks of the slowest wave / launch time, G/s");
    for (double r : rate) printf("  %7.3f", r);
    printf("\n");
    return 0;
}
