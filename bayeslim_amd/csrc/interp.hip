// interp.hip -- pixel-beam interpolation gather and its deterministic adjoint (gfx950).
//
//   fwd : out[r, p] = sum_k wgts[p,k] * m[r, inds[p,k]]                (utils.py:833-841)
//   bwd : gm[r, j]  = sum_{(p,k): inds[p,k]==j} wgts[p,k] * gout[r,p]  (scatter-add of autograd,
//         recast as a gather over a CSR inverse index -> no atomics, bitwise reproducible)
//
// HBM-bound: one lane per sky pixel (fwd) / beam pixel (bwd); the lane keeps its stencil
// (indices + weights) in registers and sweeps RT map rows per block so stencil reads are
// amortised; output stores are fully coalesced, gathers of neighbouring sky pixels land in
// the same few 128-B lines of the (L2-resident) beam map row.
#include <hip/hip_runtime.h>
#include "rime_common.h"

namespace rime {

constexpr int RT = 8;     // map rows per block

template <typename T, int NC, int NNN>
__global__ void __launch_bounds__(256)
interp_gather_kernel(const T* __restrict__ m, const int* __restrict__ inds,
                     const T* __restrict__ wgts, int R, int Npb, int P, int Nnn,
                     T* __restrict__ out, int out_stride)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const int r0 = blockIdx.y * RT;
    const int r1 = min(R, r0 + RT);
    if constexpr (NNN > 0) {
        int id[NNN];
        T w[NNN];
#pragma unroll
        for (int k = 0; k < NNN; ++k) { id[k] = inds[(size_t)p * NNN + k]; w[k] = wgts[(size_t)p * NNN + k]; }
        for (int r = r0; r < r1; ++r) {
            const T* row = m + (size_t)r * Npb * NC;
            T acc[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[c] = T(0);
#pragma unroll
            for (int k = 0; k < NNN; ++k)
#pragma unroll
                for (int c = 0; c < NC; ++c) acc[c] = tfma<T>(w[k], row[(size_t)id[k] * NC + c], acc[c]);
#pragma unroll
            for (int c = 0; c < NC; ++c) out[((size_t)r * out_stride + p) * NC + c] = acc[c];
        }
    } else {
        for (int r = r0; r < r1; ++r) {
            const T* row = m + (size_t)r * Npb * NC;
            T acc[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[c] = T(0);
            for (int k = 0; k < Nnn; ++k) {
                const int id = inds[(size_t)p * Nnn + k];
                const T w = wgts[(size_t)p * Nnn + k];
#pragma unroll
                for (int c = 0; c < NC; ++c) acc[c] = tfma<T>(w, row[(size_t)id * NC + c], acc[c]);
            }
#pragma unroll
            for (int c = 0; c < NC; ++c) out[((size_t)r * out_stride + p) * NC + c] = acc[c];
        }
    }
}

// Adjoint on TRANSPOSED buffers: goutT [P_stride][R*NC] (all rows of one sky pixel contiguous),
// gmT [Npb][R*NC].  One wave = one beam pixel j x 64 consecutive row-elements; it walks the CSR
// list of j and every contribution is a coalesced 64-lane read.  (The row-major variant read
// 4 bytes per 64-B line: 11.5 GB fetched for 0.8 GB of useful data at C4, 3.3 ms.)
template <typename T>
__global__ void __launch_bounds__(256)
interp_scatter_kernel(const T* __restrict__ goutT, const int* __restrict__ csr_ptr,
                      const int* __restrict__ csr_src, const T* __restrict__ wgts,
                      int RN, int Npb, int Nnn, T* __restrict__ gmT)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + wave;
    if (j >= Npb) return;
    const int e0 = csr_ptr[j], e1 = csr_ptr[j + 1];
    for (int r = blockIdx.y * 64 + lane; r < RN; r += gridDim.y * 64) {
        T acc = T(0);
        for (int e = e0; e < e1; ++e) {
            const int src = csr_src[e];                       // wave-uniform
            acc = tfma<T>(wgts[src], goutT[(size_t)(src / Nnn) * RN + r], acc);
        }
        gmT[(size_t)j * RN + r] = acc;
    }
}

template <typename T, int NC>
static int gather_launch(const void* m, const int* inds, const void* wgts, int R, int Npb, int P,
                         int Nnn, void* out, int out_stride, hipStream_t st)
{
    dim3 grid((P + 255) / 256, (R + RT - 1) / RT), block(256);
    const T* m_ = reinterpret_cast<const T*>(m);
    const T* w_ = reinterpret_cast<const T*>(wgts);
    T* o_ = reinterpret_cast<T*>(out);
#define RIME_G(N) hipLaunchKernelGGL((interp_gather_kernel<T, NC, N>), grid, block, 0, st, m_, inds, w_, R, Npb, P, Nnn, o_, out_stride)
    switch (Nnn) {
        case 1: RIME_G(1); break;
        case 4: RIME_G(4); break;
        case 6: RIME_G(6); break;
        case 9: RIME_G(9); break;
        case 16: RIME_G(16); break;
        default: RIME_G(0); break;
    }
#undef RIME_G
    return check_launch();
}

} // namespace rime

using namespace rime;

extern "C" int rime_interp_gather_fwd(int dtype, int is_complex, const void* m, const int* inds,
                                      const void* wgts, int R, int Npb, int P, int Nnn,
                                      void* out, int out_stride, void* stream)
{
    if (!m || !inds || !wgts || !out) return RIME_EINVAL;
    if (R <= 0 || Npb <= 0 || P <= 0 || Nnn <= 0 || out_stride < P) return RIME_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == RIME_F32)
        return is_complex ? gather_launch<float, 2>(m, inds, wgts, R, Npb, P, Nnn, out, out_stride, st)
                          : gather_launch<float, 1>(m, inds, wgts, R, Npb, P, Nnn, out, out_stride, st);
    if (dtype == RIME_F64)
        return is_complex ? gather_launch<double, 2>(m, inds, wgts, R, Npb, P, Nnn, out, out_stride, st)
                          : gather_launch<double, 1>(m, inds, wgts, R, Npb, P, Nnn, out, out_stride, st);
    return RIME_EINVAL;
}

template <typename T>
static int scatter_launch(const void* goutT, const int* csr_ptr, const int* csr_src, const void* wgts,
                          int RN, int Npb, int Nnn, void* gmT, hipStream_t st)
{
    const int ny = std::min((RN + 63) / 64, 16);
    dim3 grid((Npb + 3) / 4, ny), block(256);
    hipLaunchKernelGGL((interp_scatter_kernel<T>), grid, block, 0, st, reinterpret_cast<const T*>(goutT),
                       csr_ptr, csr_src, reinterpret_cast<const T*>(wgts), RN, Npb, Nnn,
                       reinterpret_cast<T*>(gmT));
    return check_launch();
}

extern "C" int rime_interp_scatter_bwd(int dtype, int is_complex, const void* goutT,
                                       const int* csr_ptr, const int* csr_src, const void* wgts,
                                       int R, int Npb, int P, int Nnn, void* gmT, void* stream)
{
    if (!goutT || !csr_ptr || !csr_src || !wgts || !gmT) return RIME_EINVAL;
    if (R <= 0 || Npb <= 0 || P <= 0 || Nnn <= 0) return RIME_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int RN = R * (is_complex ? 2 : 1);
    if (dtype == RIME_F32) return scatter_launch<float>(goutT, csr_ptr, csr_src, wgts, RN, Npb, Nnn, gmT, st);
    if (dtype == RIME_F64) return scatter_launch<double>(goutT, csr_ptr, csr_src, wgts, RN, Npb, Nnn, gmT, st);
    return RIME_EINVAL;
}
