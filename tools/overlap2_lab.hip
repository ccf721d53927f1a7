// overlap2_lab.hip -- same-wave MFMA + independent VALU interleave on gfx950: time vs VALU ops per MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// NV = independent v_fma_f32 per MFMA; SHAPE 0: 32x32x16 f16 (8 passes), 1: 16x16x32 f16 (4 passes)
template <int NV, int SHAPE, int WAVES>
__global__ void __launch_bounds__(64 * WAVES) k(int iters, float* out)
{
    f32x16 acc0, acc1; f32x4 c0, c1;
    for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
    for (int e = 0; e < 4; ++e) { c0[e] = 0.f; c1[e] = 0.f; }
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e * 0.5f); }
    float v[8];
    for (int e = 0; e < 8; ++e) v[e] = threadIdx.x * 0.01f + e;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (SHAPE == 0) {
                if (u & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc1, 0, 0, 0);
                else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
            } else {
                if (u & 1) c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, c1, 0, 0, 0);
                else c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < NV; ++q) v[q & 7] = fmaf(v[q & 7], 1.0001f, 0.5f);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
            if (NV > 0) __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);  // NV VALU
        }
    }
    float res = 0.f;
    for (int e = 0; e < 16; ++e) res += acc0[e] + acc1[e];
    for (int e = 0; e < 4; ++e) res += c0[e] + c1[e];
    for (int e = 0; e < 8; ++e) res += v[e];
    if (res == 123.456f) out[0] = res;
}

template <int NV, int SHAPE, int WAVES>
float run(float* d)
{
    const int iters = 20000;
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<NV, SHAPE, WAVES>), dim3(256 * 4), dim3(64 * WAVES), 0, 0, iters, d);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<NV, SHAPE, WAVES>), dim3(256 * 4), dim3(64 * WAVES), 0, 0, iters, d);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); return ms;
}

int main()
{
    float* d; CHK(hipMalloc(&d, 64));
    printf("one wave per SIMD (4-wave blocks, 4 blocks/CU worth of grid -> serial rounds)\n");
    printf("32x32x16: NV=0 %.3f  2 %.3f  4 %.3f  6 %.3f  8 %.3f  12 %.3f  16 %.3f\n",
           run<0, 0, 4>(d), run<2, 0, 4>(d), run<4, 0, 4>(d), run<6, 0, 4>(d), run<8, 0, 4>(d), run<12, 0, 4>(d), run<16, 0, 4>(d));
    printf("16x16x32: NV=0 %.3f  1 %.3f  2 %.3f  3 %.3f  4 %.3f  6 %.3f  8 %.3f\n",
           run<0, 1, 4>(d), run<1, 1, 4>(d), run<2, 1, 4>(d), run<3, 1, 4>(d), run<4, 1, 4>(d), run<6, 1, 4>(d), run<8, 1, 4>(d));
    return 0;
}
