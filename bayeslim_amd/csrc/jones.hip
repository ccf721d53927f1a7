// jones.hip -- full-polarisation beam x sky product, elementwise and fused (gfx950).
//
//   psky[a, d] = sum_{b, c} J1[a, b] S[b, c] conj(J2[d, c])          (J_p S J_q^dagger per pixel and channel)
//
// Replaces the two broadcast-multiply-sum steps of PixelBeam.apply_beam's 4-pol branch (beam_model.py:345-363: the
// einsum "ab...,bc...,dc...->ad..."), whose torch composition materialises two (2, 2, 2, ..., Nf, P) temporaries --
// 8 x the size of psky each, and again in the backward.  HBM-bound: algorithmic bytes per (model pair, channel, pixel):
// forward 4 J (16 B real / 32 B complex) [x 2 when J2 != J1] + 4 S (32 B) + 4 psky (32 B); backward the same inputs +
// 4 g (32 B) and the three gradients out.
//   J1, J2 : T or complex<T> [2, 2, N]  (N = Nmp * Nf * P, the (a, b) planes N elements apart)
//   S      : complex<T> [2, 2, Ns] with Ns dividing N (the sky is shared by the Nmp model pairs: element n reads n % Ns)
//   out, g : complex<T> [2, 2, N]
// Backward (torch's convention for complex tensors: grad = dL/dRe + i dL/dIm):
//   gS  = J1^H g J2         [2, 2, N]  (the caller sums over model pairs when Ns < N)
//   gJ1 = g (S J2^H)^H      (real part for a real beam)
//   gJ2 = g^H (J1 S)        (real part for a real beam)
#include <hip/hip_runtime.h>
#include "rime_common.h"

namespace rime {

template <typename T> struct cx { T x, y; };
template <typename T> __device__ __forceinline__ cx<T> cmul(cx<T> a, cx<T> b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
template <typename T> __device__ __forceinline__ cx<T> cmulc(cx<T> a, cx<T> b) { return {a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y}; }   // a conj(b)
template <typename T> __device__ __forceinline__ cx<T> cadd(cx<T> a, cx<T> b) { return {a.x + b.x, a.y + b.y}; }
template <typename T> __device__ __forceinline__ cx<T> cconj(cx<T> a) { return {a.x, -a.y}; }

template <typename T, bool REALB>
__device__ __forceinline__ cx<T> load_j(const T* J, size_t plane, size_t N, size_t n)
{
    if constexpr (REALB) return {J[plane * N + n], T(0)};
    else { const cx<T>* p = reinterpret_cast<const cx<T>*>(J); return p[plane * N + n]; }
}

// C = A B for 2 x 2 complex matrices stored [row * 2 + col]
template <typename T> __device__ __forceinline__ void mm(const cx<T>* A, const cx<T>* B, cx<T>* C)
{
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) C[i * 2 + j] = cadd(cmul(A[i * 2], B[j]), cmul(A[i * 2 + 1], B[2 + j]));
}
// C = A B^H
template <typename T> __device__ __forceinline__ void mmh(const cx<T>* A, const cx<T>* B, cx<T>* C)
{
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) C[i * 2 + j] = cadd(cmulc(A[i * 2], B[j * 2]), cmulc(A[i * 2 + 1], B[j * 2 + 1]));
}
// C = A^H B
template <typename T> __device__ __forceinline__ void hmm(const cx<T>* A, const cx<T>* B, cx<T>* C)
{
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) C[i * 2 + j] = cadd(cmul(cconj(A[i]), B[j]), cmul(cconj(A[2 + i]), B[2 + j]));
}

template <typename T, bool REALB>
__global__ void __launch_bounds__(256)
jones_apply_fwd_kernel(const T* __restrict__ J1, const T* __restrict__ J2, const T* __restrict__ S, size_t N, size_t Ns,
                       T* __restrict__ out)
{
    const size_t n = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const size_t ns = n % Ns;
    cx<T> a[4], b[4], s[4], m[4], o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        a[k] = load_j<T, REALB>(J1, k, N, n);
        b[k] = (J2 == J1) ? a[k] : load_j<T, REALB>(J2, k, N, n);
        s[k] = reinterpret_cast<const cx<T>*>(S)[(size_t)k * Ns + ns];
    }
    mm(a, s, m);
    mmh(m, b, o);
#pragma unroll
    for (int k = 0; k < 4; ++k) reinterpret_cast<cx<T>*>(out)[(size_t)k * N + n] = o[k];
}

template <typename T, bool REALB>
__global__ void __launch_bounds__(256)
jones_apply_bwd_kernel(const T* __restrict__ J1, const T* __restrict__ J2, const T* __restrict__ S, const T* __restrict__ G,
                       size_t N, size_t Ns, T* __restrict__ gJ1, T* __restrict__ gJ2, T* __restrict__ gS)
{
    const size_t n = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const size_t ns = n % Ns;
    cx<T> a[4], b[4], s[4], g[4], t[4], r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        a[k] = load_j<T, REALB>(J1, k, N, n);
        b[k] = (J2 == J1) ? a[k] : load_j<T, REALB>(J2, k, N, n);
        s[k] = reinterpret_cast<const cx<T>*>(S)[(size_t)k * Ns + ns];
        g[k] = reinterpret_cast<const cx<T>*>(G)[(size_t)k * N + n];
    }
    // gS = J1^H g J2
    hmm(a, g, t);
    mm(t, b, r);
#pragma unroll
    for (int k = 0; k < 4; ++k) reinterpret_cast<cx<T>*>(gS)[(size_t)k * N + n] = r[k];
    // gJ1 = g (S J2^H)^H
    mmh(s, b, t);
    mmh(g, t, r);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if constexpr (REALB) gJ1[(size_t)k * N + n] = r[k].x;
        else reinterpret_cast<cx<T>*>(gJ1)[(size_t)k * N + n] = r[k];
    }
    // gJ2 = g^H (J1 S)
    mm(a, s, t);
    hmm(g, t, r);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if constexpr (REALB) gJ2[(size_t)k * N + n] = r[k].x;
        else reinterpret_cast<cx<T>*>(gJ2)[(size_t)k * N + n] = r[k];
    }
}

// ---------------------------------------------------------------------------------------
// Stokes I + fractional polarisation -> coherency matrix, one pass each way (sky_model.py:1160-1300, the branch a
// Stokes-I sky with fractions (fQ, fU, fV) takes):
//     C = I [[1 + fQ, fU - i fV], [fU + i fV, 1 - fQ]]
// The torch composition is ~15 elementwise / stack kernels over (Nf x Npix) maps and as many in the backward (4.8 of the
// 8 ms of torch glue in a C5 rank step, profiles/r04/glue_ops_c5.txt); HBM-bound: 4 B in + 32 B out per (channel, pixel).
// Fractions are read through element strides (0 = broadcast): f[k * fs_k + r * fs_r + p * fs_p].
// Backward for the REAL input I (torch's convention, grad = dL/dRe + i dL/dIm; out = c I -> gI = Re(conj(c) g)):
//     gI = (1 + fQ) Re g00 + (1 - fQ) Re g11 + fU (Re g01 + Re g10) + fV (Im g10 - Im g01)
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256)
stokes2coh_fwd_kernel(const T* __restrict__ I, const T* __restrict__ fr, long long fs_k, long long fs_r, long long fs_p,
                      long long R, long long P, T* __restrict__ out)
{
    const long long N = R * P;
    const long long n = (long long)blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const long long r = n / P, p = n - r * P;
    const T* f = fr + r * fs_r + p * fs_p;
    const T v = I[n], fq = f[0], fu = f[fs_k], fv = f[2 * fs_k];
    cx<T>* o = reinterpret_cast<cx<T>*>(out);
    o[n] = {v + v * fq, T(0)};
    o[N + n] = {v * fu, -(v * fv)};
    o[2 * N + n] = {v * fu, v * fv};
    o[3 * N + n] = {v - v * fq, T(0)};
}

template <typename T>
__global__ void __launch_bounds__(256)
stokes2coh_bwd_kernel(const T* __restrict__ g, const T* __restrict__ fr, long long fs_k, long long fs_r, long long fs_p,
                      long long R, long long P, T* __restrict__ gI)
{
    const long long N = R * P;
    const long long n = (long long)blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const long long r = n / P, p = n - r * P;
    const T* f = fr + r * fs_r + p * fs_p;
    const T fq = f[0], fu = f[fs_k], fv = f[2 * fs_k];
    const cx<T>* gg = reinterpret_cast<const cx<T>*>(g);
    const cx<T> g00 = gg[n], g01 = gg[N + n], g10 = gg[2 * N + n], g11 = gg[3 * N + n];
    gI[n] = (g00.x + g11.x) + fq * (g00.x - g11.x) + fu * (g01.x + g10.x) + fv * (g10.y - g01.y);
}

} // namespace rime

using namespace rime;

extern "C" int rime_stokes2coh_fwd(int dtype, const void* stokesI, const void* frac, long long fs_k, long long fs_r,
                                   long long fs_p, long long R, long long P, void* coh, void* stream)
{
    if (!stokesI || !frac || !coh || R <= 0 || P <= 0 || fs_k < 0 || fs_r < 0 || fs_p < 0) return RIME_EINVAL;
    const long long N = R * P;
    if ((N + 255) / 256 > 0x7fffffffLL) return RIME_EUNSUPPORTED;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    dim3 grid((unsigned)((N + 255) / 256));
    if (dtype == RIME_F32)
        hipLaunchKernelGGL((stokes2coh_fwd_kernel<float>), grid, dim3(256), 0, st, (const float*)stokesI, (const float*)frac,
                           fs_k, fs_r, fs_p, R, P, (float*)coh);
    else if (dtype == RIME_F64)
        hipLaunchKernelGGL((stokes2coh_fwd_kernel<double>), grid, dim3(256), 0, st, (const double*)stokesI, (const double*)frac,
                           fs_k, fs_r, fs_p, R, P, (double*)coh);
    else return RIME_EINVAL;
    return check_launch();
}

extern "C" int rime_stokes2coh_bwd(int dtype, const void* gcoh, const void* frac, long long fs_k, long long fs_r,
                                   long long fs_p, long long R, long long P, void* gI, void* stream)
{
    if (!gcoh || !frac || !gI || R <= 0 || P <= 0 || fs_k < 0 || fs_r < 0 || fs_p < 0) return RIME_EINVAL;
    const long long N = R * P;
    if ((N + 255) / 256 > 0x7fffffffLL) return RIME_EUNSUPPORTED;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    dim3 grid((unsigned)((N + 255) / 256));
    if (dtype == RIME_F32)
        hipLaunchKernelGGL((stokes2coh_bwd_kernel<float>), grid, dim3(256), 0, st, (const float*)gcoh, (const float*)frac,
                           fs_k, fs_r, fs_p, R, P, (float*)gI);
    else if (dtype == RIME_F64)
        hipLaunchKernelGGL((stokes2coh_bwd_kernel<double>), grid, dim3(256), 0, st, (const double*)gcoh, (const double*)frac,
                           fs_k, fs_r, fs_p, R, P, (double*)gI);
    else return RIME_EINVAL;
    return check_launch();
}

extern "C" int rime_jones_apply_fwd(int dtype, int beam_complex, const void* J1, const void* J2, const void* S,
                                    long long N, long long Ns, void* out, void* stream)
{
    if (!J1 || !J2 || !S || !out || N <= 0 || Ns <= 0 || N % Ns != 0) return RIME_EINVAL;
    if ((N + 255) / 256 > 0x7fffffffLL) return RIME_EUNSUPPORTED;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    dim3 grid((unsigned)((N + 255) / 256));
#define RIME_J(T, RB) hipLaunchKernelGGL((jones_apply_fwd_kernel<T, RB>), grid, dim3(256), 0, st, (const T*)J1, (const T*)J2, \
                                         (const T*)S, (size_t)N, (size_t)Ns, (T*)out)
    if (dtype == RIME_F32) { if (beam_complex) RIME_J(float, false); else RIME_J(float, true); }
    else if (dtype == RIME_F64) { if (beam_complex) RIME_J(double, false); else RIME_J(double, true); }
    else return RIME_EINVAL;
#undef RIME_J
    return check_launch();
}

extern "C" int rime_jones_apply_bwd(int dtype, int beam_complex, const void* J1, const void* J2, const void* S,
                                    const void* G, long long N, long long Ns, void* gJ1, void* gJ2, void* gS, void* stream)
{
    if (!J1 || !J2 || !S || !G || !gJ1 || !gJ2 || !gS || N <= 0 || Ns <= 0 || N % Ns != 0) return RIME_EINVAL;
    if ((N + 255) / 256 > 0x7fffffffLL) return RIME_EUNSUPPORTED;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    dim3 grid((unsigned)((N + 255) / 256));
#define RIME_J(T, RB) hipLaunchKernelGGL((jones_apply_bwd_kernel<T, RB>), grid, dim3(256), 0, st, (const T*)J1, (const T*)J2, \
                                         (const T*)S, (const T*)G, (size_t)N, (size_t)Ns, (T*)gJ1, (T*)gJ2, (T*)gS)
    if (dtype == RIME_F32) { if (beam_complex) RIME_J(float, false); else RIME_J(float, true); }
    else if (dtype == RIME_F64) { if (beam_complex) RIME_J(double, false); else RIME_J(double, true); }
    else return RIME_EINVAL;
#undef RIME_J
    return check_launch();
}
