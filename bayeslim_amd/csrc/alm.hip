// alm.hip -- a_lm -> pixel transform and its adjoint (gfx950).
//
//   fwd : out[r, j]  = sum_c ( are[r,c] Yre[c,j] - aim[r,c] Yim[c,j] ) = Re( alm @ Ylm )
//   bwd : galm[r, c] = sum_j gout[r, j] * conj(Ylm[c, j])
// Replaces AlmModel.forward_alm (sph_harm.py:1342-1372, real_output=True) and the einsum
// backward autograd derives for it.
//
// Both are real GEMMs against the interleaved complex Ylm [Ncoeff, Npix, 2] viewed as a
// (2 Ncoeff) x Npix matrix (fwd: K = 2 Ncoeff) or its transpose (bwd: K = Npix).  Ylm is the
// only large operand (C3: 3.3 GB) and is streamed exactly once per <= BM rows of alm; the
// alm / gout operand is small and stays in L2.
//
// v1 (this file): f32/f64 VALU register-tiled kernels -- lanes along the contiguous pixel
// axis, BM rows of accumulators per lane, the alm operand read as wave-uniform scalars.
#include <hip/hip_runtime.h>
#include "rime_common.h"

namespace rime {

// ---------------------------------------------------------------------------------------
// forward: lane = pixel, RTA rows per block (grid.y walks row blocks)
// ---------------------------------------------------------------------------------------
template <typename T, int RTA>
__global__ void __launch_bounds__(256)
alm2pix_fwd_kernel(const T* __restrict__ alm, const T* __restrict__ Ylm, int R, int Ncoeff,
                   int Npix, T* __restrict__ out)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int r0 = blockIdx.y * RTA;
    const int jc = min(j, Npix - 1);
    T acc[RTA];
#pragma unroll
    for (int i = 0; i < RTA; ++i) acc[i] = T(0);
    for (int c = 0; c < Ncoeff; ++c) {
        const T yre = Ylm[((size_t)c * Npix + jc) * 2];
        const T yim = Ylm[((size_t)c * Npix + jc) * 2 + 1];
#pragma unroll
        for (int i = 0; i < RTA; ++i) {
            const int r = min(r0 + i, R - 1);                  // wave-uniform -> scalar loads
            const T are = alm[((size_t)r * Ncoeff + c) * 2];
            const T aim = alm[((size_t)r * Ncoeff + c) * 2 + 1];
            acc[i] = tfma<T>(are, yre, acc[i]);
            acc[i] = tfma<T>(-aim, yim, acc[i]);
        }
    }
    if (j < Npix) {
#pragma unroll
        for (int i = 0; i < RTA; ++i)
            if (r0 + i < R) out[(size_t)(r0 + i) * Npix + j] = acc[i];
    }
}

// ---------------------------------------------------------------------------------------
// backward: block = CT coefficients x RTB rows; lanes stride over pixels; deterministic
// wave-shuffle + LDS reduction at the end
// ---------------------------------------------------------------------------------------
template <typename T, int CT, int RTB>
__global__ void __launch_bounds__(256)
alm2pix_bwd_kernel(const T* __restrict__ gout, const T* __restrict__ Ylm, int R, int Ncoeff,
                   int Npix, T* __restrict__ galm)
{
    __shared__ T red[4][CT * RTB * 2];
    const int c0 = blockIdx.x * CT;
    const int r0 = blockIdx.y * RTB;
    T acc[CT][RTB][2];
#pragma unroll
    for (int a = 0; a < CT; ++a)
#pragma unroll
        for (int i = 0; i < RTB; ++i) { acc[a][i][0] = T(0); acc[a][i][1] = T(0); }
    for (int j = threadIdx.x; j < Npix; j += blockDim.x) {
        T g[RTB];
#pragma unroll
        for (int i = 0; i < RTB; ++i) g[i] = gout[(size_t)min(r0 + i, R - 1) * Npix + j];
#pragma unroll
        for (int a = 0; a < CT; ++a) {
            const int c = min(c0 + a, Ncoeff - 1);
            const T yre = Ylm[((size_t)c * Npix + j) * 2];
            const T yim = Ylm[((size_t)c * Npix + j) * 2 + 1];
#pragma unroll
            for (int i = 0; i < RTB; ++i) {
                acc[a][i][0] = tfma<T>(g[i], yre, acc[a][i][0]);
                acc[a][i][1] = tfma<T>(-g[i], yim, acc[a][i][1]);
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int a = 0; a < CT; ++a)
#pragma unroll
        for (int i = 0; i < RTB; ++i)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                T v = acc[a][i][q];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
                if (lane == 0) red[wave][(a * RTB + i) * 2 + q] = v;
            }
    __syncthreads();
    const int nw = blockDim.x >> 6;
    for (int e = threadIdx.x; e < CT * RTB * 2; e += blockDim.x) {
        T v = T(0);
        for (int w = 0; w < nw; ++w) v += red[w][e];
        const int q = e & 1, i = (e >> 1) % RTB, a = (e >> 1) / RTB;
        if (c0 + a < Ncoeff && r0 + i < R)
            galm[((size_t)(r0 + i) * Ncoeff + c0 + a) * 2 + q] = v;
    }
}

} // namespace rime

using namespace rime;

extern "C" int rime_alm2pix_fwd(int dtype, const void* alm, const void* Ylm, int R, int Ncoeff,
                                int Npix, void* out, void* stream)
{
    if (!alm || !Ylm || !out || R <= 0 || Ncoeff <= 0 || Npix <= 0) return RIME_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == RIME_F32) {
        constexpr int RTA = 32;
        dim3 grid((Npix + 255) / 256, (R + RTA - 1) / RTA);
        hipLaunchKernelGGL((alm2pix_fwd_kernel<float, RTA>), grid, dim3(256), 0, st,
                           (const float*)alm, (const float*)Ylm, R, Ncoeff, Npix, (float*)out);
    } else if (dtype == RIME_F64) {
        constexpr int RTA = 16;
        dim3 grid((Npix + 255) / 256, (R + RTA - 1) / RTA);
        hipLaunchKernelGGL((alm2pix_fwd_kernel<double, RTA>), grid, dim3(256), 0, st,
                           (const double*)alm, (const double*)Ylm, R, Ncoeff, Npix, (double*)out);
    } else return RIME_EINVAL;
    return check_launch();
}

extern "C" int rime_alm2pix_bwd(int dtype, const void* gout, const void* Ylm, int R, int Ncoeff,
                                int Npix, void* galm, void* stream)
{
    if (!gout || !Ylm || !galm || R <= 0 || Ncoeff <= 0 || Npix <= 0) return RIME_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    constexpr int CT = 4, RTB = 8;
    dim3 grid((Ncoeff + CT - 1) / CT, (R + RTB - 1) / RTB);
    if (dtype == RIME_F32)
        hipLaunchKernelGGL((alm2pix_bwd_kernel<float, CT, RTB>), grid, dim3(256), 0, st,
                           (const float*)gout, (const float*)Ylm, R, Ncoeff, Npix, (float*)galm);
    else if (dtype == RIME_F64)
        hipLaunchKernelGGL((alm2pix_bwd_kernel<double, CT, RTB>), grid, dim3(256), 0, st,
                           (const double*)gout, (const double*)Ylm, R, Ncoeff, Npix, (double*)galm);
    else return RIME_EINVAL;
    return check_launch();
}
