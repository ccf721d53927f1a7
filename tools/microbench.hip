// microbench.hip -- gfx950 instruction-rate and trig-accuracy probes that shaped the fringe
// kernel (DESIGN.md "measured constants").  Build: hipcc -O3 --offload-arch=gfx950 microbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ void __launch_bounds__(256) rate_kernel(float* out, int iters)
{
    float s = threadIdx.x * 1e-6f;
    if constexpr (MODE == 0) {            // v_fma_f32, 8 independent chains
        float a[8];
        for (int i = 0; i < 8; ++i) a[i] = s + i;
        float b = 0.999f, c = 1e-3f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        }
        float t = 0; for (int i = 0; i < 8; ++i) t += a[i];
        out[blockIdx.x * blockDim.x + threadIdx.x] = t;
    } else if constexpr (MODE == 1) {     // v_pk_fma_f32
        f2 a[8];
        for (int i = 0; i < 8; ++i) a[i] = f2{s + i, s - i};
        f2 b = {0.999f, 0.998f}, c = {1e-3f, 2e-3f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        }
        float t = 0; for (int i = 0; i < 8; ++i) t += a[i].x + a[i].y;
        out[blockIdx.x * blockDim.x + threadIdx.x] = t;
    } else if constexpr (MODE == 2) {     // v_fma_f64
        double a[8];
        for (int i = 0; i < 8; ++i) a[i] = s + i;
        double b = 0.999, c = 1e-3;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f64 %0, %1, %0, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        }
        double t = 0; for (int i = 0; i < 8; ++i) t += a[i];
        out[blockIdx.x * blockDim.x + threadIdx.x] = (float)t;
    } else if constexpr (MODE == 3) {     // v_sin_f32
        float a[8];
        for (int i = 0; i < 8; ++i) a[i] = 0.01f * (s + i);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_sin_f32 %0, %0" : "+v"(a[i]));
        }
        float t = 0; for (int i = 0; i < 8; ++i) t += a[i];
        out[blockIdx.x * blockDim.x + threadIdx.x] = t;
    } else if constexpr (MODE == 4) {     // dependent fma chain (latency) single chain
        float a = s, b = 0.999f, c = 1e-3f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 64; ++r) asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(a) : "v"(b), "v"(c));
        }
        out[blockIdx.x * blockDim.x + threadIdx.x] = a;
    }
}

__device__ __forceinline__ void sincos_turns(float r, float& s, float& c)
{
    float q = rintf(4.0f * r);
    float t = fmaf(q, -0.25f, r);
    float u = t * t;
    float ps = fmaf(u, -75.40161269908938f, 81.59254287120774f);
    ps = fmaf(u, ps, -41.34166257054729f);
    ps = fmaf(u, ps, 6.283185287812946f);
    float s0 = ps * t;
    float pc = fmaf(u, 59.220168979635126f, -85.44284467368314f);
    pc = fmaf(u, pc, 64.93931613571324f);
    pc = fmaf(u, pc, -19.7392086501306f);
    float c0 = fmaf(u, pc, 1.0f);
    int qi = (int)q;
    bool swap = qi & 1;
    float cc = swap ? s0 : c0;
    float ss = swap ? c0 : s0;
    uint32_t sgn_c = ((uint32_t)(qi + 1) & 2u) << 30;
    uint32_t sgn_s = ((uint32_t)qi & 2u) << 30;
    c = __uint_as_float(__float_as_uint(cc) ^ sgn_c);
    s = __uint_as_float(__float_as_uint(ss) ^ sgn_s);
}

__global__ void trig_kernel(const float* x, float* o, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float r = x[i];
    o[i] = __builtin_amdgcn_sinf(r);          // v_sin_f32 (input in turns)
    o[n + i] = __builtin_amdgcn_cosf(r);
    float s, c;
    sincos_turns(r, s, c);
    o[2 * n + i] = s; o[3 * n + i] = c;
    float s2, c2;
    sincospif(2.0f * r, &s2, &c2);
    o[4 * n + i] = s2; o[5 * n + i] = c2;
}

template <int MODE>
static int run_rate(const char* name, double flop_per_inst_lane, int blocks_per_cu)
{
    int iters = 2000;
    int nb = 256 * blocks_per_cu;
    float* out; CHK(hipMalloc(&out, (size_t)nb * 256 * 4));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(nb), dim3(256), 0, 0, out, 10);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(nb), dim3(256), 0, 0, out, iters);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    double insts = (double)nb * 256 * iters * 64.0;            // lane-instructions
    printf("%-22s blocks/CU=%d  %.3f ms  %.2f Tlane-inst/s  %.1f TFLOP/s\n", name, blocks_per_cu, ms,
           insts / ms * 1e-9, insts * flop_per_inst_lane / ms * 1e-9);
    CHK(hipFree(out));
    return 0;
}

int main()
{
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    printf("device %s  CUs %d  clock %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    for (int bpc : {1, 2, 4, 8}) {
        if (run_rate<0>("v_fma_f32", 2, bpc)) return 1;
        if (run_rate<1>("v_pk_fma_f32", 4, bpc)) return 1;
    }
    if (run_rate<2>("v_fma_f64", 2, 4)) return 1;
    if (run_rate<3>("v_sin_f32", 1, 4)) return 1;
    if (run_rate<4>("v_fma_f32 dep-chain", 2, 1)) return 1;
    if (run_rate<4>("v_fma_f32 dep-chain", 2, 2)) return 1;
    if (run_rate<4>("v_fma_f32 dep-chain", 2, 4)) return 1;

    const int n = 1 << 20;
    std::vector<float> x(n), o(6 * n);
    for (int i = 0; i < n; ++i) x[i] = -0.5f + (float)i / (float)(n - 1);
    float *dx, *dout; CHK(hipMalloc(&dx, n * 4)); CHK(hipMalloc(&dout, 6 * n * 4));
    CHK(hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(trig_kernel, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
    CHK(hipMemcpy(o.data(), dout, 6 * n * 4, hipMemcpyDeviceToHost));
    const char* names[3] = {"v_sin/v_cos_f32 (hw)", "sincos_turns (poly)", "sincospif (ocml)"};
    for (int v = 0; v < 3; ++v) {
        double es = 0, ec = 0;
        for (int i = 0; i < n; ++i) {
            double a = 2.0 * M_PI * (double)x[i];
            es = fmax(es, fabs(o[(2 * v) * n + i] - sin(a)));
            ec = fmax(ec, fabs(o[(2 * v + 1) * n + i] - cos(a)));
        }
        printf("%-24s max|err| sin %.3e  cos %.3e\n", names[v], es, ec);
    }
    return 0;
}
