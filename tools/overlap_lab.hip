// overlap_lab.hip -- do VALU work and MFMA work of DIFFERENT waves on one SIMD overlap on gfx950?
// Block = 512 threads (2 waves per SIMD).  Per VALU flavour: all waves MFMA; all waves VALU
// (calibrated to the same duration); waves 0-3 MFMA + waves 4-7 VALU.  Perfect overlap: both = max.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int KIND>
__device__ __forceinline__ float valu_work(int iters, float seed)
{
    float a = seed, b = seed * 0.5f, c = seed + 1.f, d = seed - 1.f;
    double x = seed;
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) { a = fmaf(a, 1.0001f, 0.5f); b = fmaf(b, 0.9999f, 0.25f); c = fmaf(c, 1.0002f, 0.1f); d = fmaf(d, 0.9998f, 0.2f); }
        } else if (KIND == 1) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { a = __builtin_amdgcn_sinf(a) + 0.3f; b = __builtin_amdgcn_cosf(b) + 0.2f; c = fmaf(c, 1.0002f, 0.1f); d = fmaf(d, 0.9998f, 0.2f); }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) { x = fma(x, 1.0000001, 0.5); a = fmaf(a, 1.0001f, (float)x); b = fmaf(b, 0.9999f, 0.25f); }
        }
    }
    return a + b + c + d + (float)x;
}

template <int KIND>
__global__ void __launch_bounds__(512) k(int mode, int it_m, int it_v, float* out)
{
#if defined(LAB_PRIO)
    if ((threadIdx.x >> 6) < 4) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(3);
#endif
    const int wave = threadIdx.x >> 6;
    const bool do_m = (mode == 0) || ((mode == 2 || mode == 3) && wave < 4);
    const bool do_v = (mode == 1) || ((mode == 2 || mode == 4) && wave >= 4);
    float res = 0.f;
    const long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    if (do_m) {
        f32x16 acc0, acc1;
        for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
        f16x8 a, b;
        for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e * 0.5f); }
        for (int i = 0; i < it_m; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
#if defined(LAB_NOP)
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_nop 7\n s_nop 7\n s_nop 7");
                __builtin_amdgcn_sched_barrier(0);
#endif
#if defined(LAB_SLEEP)
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_sleep(LAB_SLEEP);
                __builtin_amdgcn_sched_barrier(0);
#endif
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc1, 0, 0, 0);
#if defined(LAB_NOP)
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_nop 7\n s_nop 7\n s_nop 7");
                __builtin_amdgcn_sched_barrier(0);
#endif
#if defined(LAB_SLEEP)
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_sleep(LAB_SLEEP);
                __builtin_amdgcn_sched_barrier(0);
#endif
            }
        }
        for (int e = 0; e < 16; ++e) res += acc0[e] + acc1[e];
    }
    if (do_v) res += valu_work<KIND>(it_v, threadIdx.x * 1e-3f);
    const long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    if (blockIdx.x == 7 && (threadIdx.x & 63) == 0) { out[2 + 2 * wave] = (float)(c1 - c0); out[3 + 2 * wave] = (float)(w1 - w0); }
    if (res == 123.456f) out[0] = res;
}

template <int KIND>
float run(int mode, int it_m, int it_v, float* d)
{
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<KIND>, dim3(256 * 4), dim3(512), 0, 0, mode, it_m, it_v, d);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<KIND>, dim3(256 * 4), dim3(512), 0, 0, mode, it_m, it_v, d);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    float h[20]; CHK(hipMemcpy(h, d, 80, hipMemcpyDeviceToHost));
    printf("   [mode %d: wave0 %.0f cyc / %.0f ticks(100MHz) = %.0f MHz; wave4 %.0f cyc / %.0f ticks]\n", mode, h[2], h[3], h[2] / h[3] * 100.0, h[10], h[11]);
    return ms;
}

template <int KIND>
void trial(const char* name, float* d)
{
    const int it_m = 20000;
    float m = run<KIND>(0, it_m, 0, d);
    int itv = 20000;
    float v = run<KIND>(1, 0, itv, d); itv = (int)(itv * m / v); v = run<KIND>(1, 0, itv, d);
    float b = run<KIND>(2, it_m, itv, d);
    float hm = run<KIND>(3, it_m, itv, d), hv = run<KIND>(4, it_m, itv, d);
    printf("%-6s %8.3f  %8.3f | half waves: mfma %8.3f  valu %8.3f  both %8.3f\n", name, m, v, hm, hv, b);
}

int main()
{
    float* d; CHK(hipMalloc(&d, 256));
    printf("kind   mfma_all  valu_all\n");
    trial<0>("fma", d); trial<1>("trans", d); trial<2>("f64", d);
    return 0;
}
