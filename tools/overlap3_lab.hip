// overlap3_lab.hip -- same-wave MFMA + independent VALU of different KINDS (gfx950): which vector instructions issue in the
// shadow of a v_mfma_f32_32x32x16_f16 of the same wave?  KIND 0 v_fma_f32, 1 v_fma_f64, 2 v_sin_f32 (transcendental),
// 3 v_cvt_pkrtz_f16_f32, 4 v_fract_f64 + v_cvt_f32_f64, 5 v_accvgpr_read.  One wave per SIMD; NV instructions per MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int NV, int KIND, bool MFMA>
__global__ void __launch_bounds__(256) k(int iters, float* out)
{
    f32x16 acc0, acc1;
    for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e * 0.5f); }
    float v[8]; double w[8]; unsigned u[8];
    for (int e = 0; e < 8; ++e) { v[e] = threadIdx.x * 0.01f + e; w[e] = threadIdx.x * 0.01 + e; u[e] = threadIdx.x + e; }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (MFMA) {
                if (m & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc1, 0, 0, 0);
                else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                const int s = q & 7;
                if (KIND == 0) v[s] = fmaf(v[s], 1.0001f, 0.5f);
                else if (KIND == 1) w[s] = fma(w[s], 1.0001, 0.5);
                else if (KIND == 2) v[s] = __builtin_amdgcn_sinf(v[s]);
                else if (KIND == 3) { auto hh = __builtin_amdgcn_cvt_pkrtz(v[s], v[(s + 1) & 7]); u[s] ^= __builtin_bit_cast(unsigned, hh); }
                else if (KIND == 4) { if (q & 1) v[s] = (float)w[s]; else w[s] = __builtin_amdgcn_fract(w[s]) + 1.5; }
            }
            if (MFMA) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (NV > 0) __builtin_amdgcn_sched_group_barrier(0x002, NV * (KIND == 3 ? 2 : 1), 0);
        }
    }
    float res = 0.f;
    for (int e = 0; e < 16; ++e) res += acc0[e] + acc1[e];
    for (int e = 0; e < 8; ++e) res += v[e] + (float)w[e] + (float)u[e];
    if (res == 123.456f) out[0] = res;
}

template <int NV, int KIND, bool MFMA>
float run(float* d)
{
    const int iters = 20000;
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<NV, KIND, MFMA>), dim3(256), dim3(256), 0, 0, iters, d);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<NV, KIND, MFMA>), dim3(256), dim3(256), 0, 0, iters, d);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); return ms;
}

template <int KIND> void row(const char* name, float* d)
{
    printf("%-22s MFMA + NV: 0 %.3f | 2 %.3f | 4 %.3f | 8 %.3f      VALU alone NV: 2 %.3f | 4 %.3f | 8 %.3f\n", name,
           run<0, KIND, true>(d), run<2, KIND, true>(d), run<4, KIND, true>(d), run<8, KIND, true>(d),
           run<2, KIND, false>(d), run<4, KIND, false>(d), run<8, KIND, false>(d));
}

int main()
{
    float* d; CHK(hipMalloc(&d, 64));
    printf("one wave per SIMD (256 blocks x 4 waves), 160 000 MFMAs per wave; ms\n");
    row<0>("v_fma_f32", d); row<1>("v_fma_f64", d); row<2>("v_sin_f32", d); row<3>("v_cvt_pkrtz + v_xor", d); row<4>("v_fract_f64 / v_cvt", d);
    return 0;
}
