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
// float32: exact-f32 matrix cores (v_mfma_f32_32x32x2_f32; one complex coefficient / one pixel
// pair per K-step), operands staged through LDS in fragment order, Ylm streamed once.
// float64 (parity oracle precision) and tiny shapes: VALU register-tiled kernels.
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


// ---------------------------------------------------------------------------------------
// f32 MFMA kernels.  v_mfma_f32_32x32x2_f32: D[32x32] += A[32x2] * B[2x32]; lane l holds
// A[i = l&31][k = l>>5], B[k = l>>5][j = l&31]; D[row = (reg&3) + 8*(reg>>2) + 4*(l>>5)][col = l&31].
// ---------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));

// forward: out[r, j] = sum_c are[r,c] Yre[c,j] - aim[r,c] Yim[c,j].
//   K-step = one coefficient c (k = re / im).  Block = 4 waves x 32 pixels; every wave keeps
//   MT row tiles (MT*32 rows) of accumulators.  alm chunks [CT coeffs][re,-im][rows] are staged in
//   LDS (row-fastest, +1 pad: conflict-free both ways); Ylm fragments come straight from global
//   memory: the 64 lanes of a wave read 64 consecutive floats (32 pixels x re/im) per coefficient.
template <int MT>
__global__ void __launch_bounds__(256)
alm2pix_fwd_mfma_kernel(const float* __restrict__ alm, const float* __restrict__ Ylm, int R,
                        int Ncoeff, int Npix, float* __restrict__ out)
{
    constexpr int CT = 32;                 // coefficients per LDS chunk
    constexpr int ROWS = MT * 32;
    constexpr int RP = ROWS + 1;           // padded row count
    __shared__ float a_lds[CT * 2 * RP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r0 = blockIdx.y * ROWS;
    const int j0 = (blockIdx.x * 4 + wave) * 32;
    const int jl = min(j0 + (lane & 31), Npix - 1);
    const int ri = lane >> 5;
    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[m][e] = 0.f;

    // Ylm fragments of one chunk live in registers and are fetched one chunk ahead, so the
    // 32 independent 256-B reads of chunk k+1 are in flight under the MFMAs of chunk k
    float bcur[CT], bnxt[CT];
    auto load_b = [&](int c0, float (&b)[CT]) {
#pragma unroll
        for (int cc = 0; cc < CT; ++cc) {
            const int c = min(c0 + cc, Ncoeff - 1);
            b[cc] = Ylm[((size_t)c * Npix + jl) * 2 + ri];
        }
    };
    load_b(0, bcur);
    for (int c0 = 0; c0 < Ncoeff; c0 += CT) {
        __syncthreads();
        // stage alm[r0 .. r0+ROWS, c0 .. c0+CT] -> a_lds[(cc*2+q)*RP + row], imaginary part negated
        // (coefficients past Ncoeff are zero-filled, so the clamped Ylm prefetch contributes 0)
        for (int i = tid; i < ROWS * CT * 2; i += 256) {
            const int e = i % (CT * 2), row = i / (CT * 2);
            const int c = c0 + (e >> 1), r = r0 + row;
            float v = 0.f;
            if (c < Ncoeff && r < R) v = alm[((size_t)r * Ncoeff + c) * 2 + (e & 1)];
            a_lds[e * RP + row] = (e & 1) ? -v : v;
        }
        if (c0 + CT < Ncoeff) load_b(c0 + CT, bnxt);
        __syncthreads();
#pragma unroll
        for (int cc = 0; cc < CT; ++cc) {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const float a = a_lds[(cc * 2 + ri) * RP + m * 32 + (lane & 31)];
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bcur[cc], acc[m], 0, 0, 0);
            }
        }
#pragma unroll
        for (int cc = 0; cc < CT; ++cc) bcur[cc] = bnxt[cc];
    }
    const int col = j0 + (lane & 31);
    if (col < Npix) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = r0 + m * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (row < R) out[(size_t)row * Npix + col] = acc[m][e];
            }
    }
}

// backward: galm[r, (c,q)] = sum_j gout[r, j] * Y[c, j, q] * (q ? -1 : +1).
//   K-step = two pixels.  Block = one tile of 16 coefficients (32 (c,q) columns) x MT*32 rows,
//   wave m owns row tile m; grid.y splits the pixel axis (partials in `part`, reduced afterwards).
//   gout and Ylm tiles of KT pixels are staged in LDS: gout [row][KT+1] (A fragments: lanes =
//   rows, odd stride), Ylm [coef][KT*2+2] (B fragments: lanes = (c,q), stride = 2 mod 32).
template <int MT>
__global__ void __launch_bounds__(64 * MT)
alm2pix_bwd_mfma_kernel(const float* __restrict__ gout, const float* __restrict__ Ylm, int R,
                        int Ncoeff, int Npix, int pix_per_split, float* __restrict__ part)
{
    constexpr int KT = 64;
    constexpr int ROWS = MT * 32;
    constexpr int GP = KT + 1;
    constexpr int YP = KT * 2 + 2;
    __shared__ float g_lds[ROWS * GP];
    __shared__ float y_lds[16 * YP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nthr = 64 * MT;
    const int c0 = blockIdx.x * 16;
    const int r0 = blockIdx.z * ROWS;
    const int jbeg = blockIdx.y * pix_per_split;
    const int jend = min(Npix, jbeg + pix_per_split);
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const int n = lane & 31, kk = lane >> 5;
    // global -> registers -> LDS, one tile ahead: the loads of tile t+1 are issued before the
    // MFMAs of tile t and written to LDS after them (single LDS buffer, two barriers per tile)
    constexpr int NG = (ROWS * KT) / (64 * MT);        // gout values per thread per tile (= 32)
    constexpr int NY = (16 * KT * 2) / (64 * MT);      // Ylm values per thread per tile
    float gq[NG], yq[NY];
    auto fetch = [&](int jt) {
#pragma unroll
        for (int u = 0; u < NG; ++u) {
            const int i = tid + u * nthr;
            const int jj = i % KT, row = i / KT;
            const int r = r0 + row, j = jt + jj;
            gq[u] = (r < R && j < jend) ? gout[(size_t)r * Npix + j] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < NY; ++u) {
            const int i = tid + u * nthr;
            const int e = i % (KT * 2), cc = i / (KT * 2);
            const int c = c0 + cc, j = jt + (e >> 1);
            const float v = (c < Ncoeff && j < jend) ? Ylm[((size_t)c * Npix + jt) * 2 + e] : 0.f;
            yq[u] = (e & 1) ? -v : v;
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int u = 0; u < NG; ++u) {
            const int i = tid + u * nthr;
            g_lds[(i / KT) * GP + (i % KT)] = gq[u];
        }
#pragma unroll
        for (int u = 0; u < NY; ++u) {
            const int i = tid + u * nthr;
            y_lds[(i / (KT * 2)) * YP + (i % (KT * 2))] = yq[u];
        }
    };
    if (jbeg < jend) fetch(jbeg);
    for (int jt = jbeg; jt < jend; jt += KT) {
        __syncthreads();                     // previous tile fully consumed
        commit();
        __syncthreads();
        if (jt + KT < jend) fetch(jt + KT);
#pragma unroll 8
        for (int jj = 0; jj < KT; jj += 2) {
            const float a = g_lds[(wave * 32 + n) * GP + jj + kk];
            const float b = y_lds[(n >> 1) * YP + (jj + kk) * 2 + (n & 1)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    }
    // D[row][col = (c,q)] -> part[split][r][c][q]
    const int c = c0 + (n >> 1);
    if (c < Ncoeff) {
        float* dst = part + (size_t)blockIdx.y * R * Ncoeff * 2;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = r0 + wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
            if (row < R) dst[((size_t)row * Ncoeff + c) * 2 + (n & 1)] = acc[e];
        }
    }
}

__global__ void alm_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, size_t len, int S)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += (size_t)gridDim.x * blockDim.x) {
        float v = 0.f;
        for (int s = 0; s < S; ++s) v += part[(size_t)s * len + i];
        out[i] = v;
    }
}

static int bwd_splits(int R, int Ncoeff, int Npix)
{
    // enough blocks to keep Ylm streaming: >= ~2048 waves; each split covers a multiple of 64 pixels
    const int MT = R > 64 ? 4 : (R > 32 ? 2 : 1);
    const long blocks = (long)((Ncoeff + 15) / 16) * ((R + MT * 32 - 1) / (MT * 32));
    long S = (2048 + blocks * MT - 1) / (blocks * MT);
    const long maxS = std::max(1, Npix / 1024);
    if (S > maxS) S = maxS;
    return (int)std::max<long>(1, S);
}

} // namespace rime

using namespace rime;

extern "C" int rime_alm2pix_fwd(int dtype, const void* alm, const void* Ylm, int R, int Ncoeff,
                                int Npix, void* out, void* stream)
{
    if (!alm || !Ylm || !out || R <= 0 || Ncoeff <= 0 || Npix <= 0) return RIME_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == RIME_F32) {
        const float* a = (const float*)alm; const float* Y = (const float*)Ylm; float* o = (float*)out;
        const int MT = R > 64 ? 4 : (R > 32 ? 2 : 1);
        dim3 grid((Npix + 127) / 128, (R + MT * 32 - 1) / (MT * 32));
        if (MT == 4) hipLaunchKernelGGL((alm2pix_fwd_mfma_kernel<4>), grid, dim3(256), 0, st, a, Y, R, Ncoeff, Npix, o);
        else if (MT == 2) hipLaunchKernelGGL((alm2pix_fwd_mfma_kernel<2>), grid, dim3(256), 0, st, a, Y, R, Ncoeff, Npix, o);
        else hipLaunchKernelGGL((alm2pix_fwd_mfma_kernel<1>), grid, dim3(256), 0, st, a, Y, R, Ncoeff, Npix, o);
    } else if (dtype == RIME_F64) {
        constexpr int RTA = 16;
        dim3 grid((Npix + 255) / 256, (R + RTA - 1) / RTA);
        hipLaunchKernelGGL((alm2pix_fwd_kernel<double, RTA>), grid, dim3(256), 0, st,
                           (const double*)alm, (const double*)Ylm, R, Ncoeff, Npix, (double*)out);
    } else return RIME_EINVAL;
    return check_launch();
}

extern "C" size_t rime_alm2pix_bwd_workspace(int dtype, int R, int Ncoeff, int Npix)
{
    if (dtype != RIME_F32) return 0;
    const int S = bwd_splits(R, Ncoeff, Npix);
    return S <= 1 ? 0 : (size_t)S * R * Ncoeff * 2 * sizeof(float);
}

extern "C" int rime_alm2pix_bwd(int dtype, const void* gout, const void* Ylm, int R, int Ncoeff,
                                int Npix, void* galm, void* workspace, size_t workspace_bytes,
                                void* stream)
{
    if (!gout || !Ylm || !galm || R <= 0 || Ncoeff <= 0 || Npix <= 0) return RIME_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == RIME_F32) {
        const int S = bwd_splits(R, Ncoeff, Npix);
        const size_t len = (size_t)R * Ncoeff * 2;
        if (S > 1 && (!workspace || workspace_bytes < (size_t)S * len * sizeof(float))) return RIME_EWORKSPACE;
        float* part = S > 1 ? (float*)workspace : (float*)galm;
        int pps = (Npix + S - 1) / S;
        pps = ((pps + 63) / 64) * 64;
        const int MT = R > 64 ? 4 : (R > 32 ? 2 : 1);
        dim3 grid((Ncoeff + 15) / 16, S, (R + MT * 32 - 1) / (MT * 32));
        const float* g = (const float*)gout; const float* Y = (const float*)Ylm;
        if (MT == 4) hipLaunchKernelGGL((alm2pix_bwd_mfma_kernel<4>), grid, dim3(256), 0, st, g, Y, R, Ncoeff, Npix, pps, part);
        else if (MT == 2) hipLaunchKernelGGL((alm2pix_bwd_mfma_kernel<2>), grid, dim3(128), 0, st, g, Y, R, Ncoeff, Npix, pps, part);
        else hipLaunchKernelGGL((alm2pix_bwd_mfma_kernel<1>), grid, dim3(64), 0, st, g, Y, R, Ncoeff, Npix, pps, part);
        if (S > 1) {
            int nb = (int)std::min<size_t>((len + 255) / 256, 2048);
            hipLaunchKernelGGL(alm_reduce_kernel, dim3(nb), dim3(256), 0, st, part, (float*)galm, len, S);
        }
        return check_launch();
    }
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
