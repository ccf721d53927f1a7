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
    RIME_MFMA_SETTLE();
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
    RIME_MFMA_SETTLE();
    if (c < Ncoeff) {
        float* dst = part + (size_t)blockIdx.y * R * Ncoeff * 2;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = r0 + wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
            if (row < R) dst[((size_t)row * Ncoeff + c) * 2 + (n & 1)] = acc[e];
        }
    }
}

// ---------------------------------------------------------------------------------------
// float32 fast path: f16 hi/lo split operands on v_mfma_f32_32x32x16_f16 (three cross products,
// f32 accumulation -- the scheme of fringe_mfma.hip; 22 significant bits, 16x the f32-MFMA rate), so
// that both directions run at the speed Ylm streams from HBM instead of the f32 matrix-core rate.
// The small operand (alm rows / gout rows) is scaled per row by a power of two, split once by
// split_rows_kernel into A-fragment order in a workspace and copied to LDS by the GEMM blocks; the
// streamed operand (Ylm, scaled by the caller's power of two y_scale) is split in registers by the
// lane that loaded it.
// ---------------------------------------------------------------------------------------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split2h(float a, float b, uint32_t& hi, uint32_t& lo)
{
    auto h = __builtin_amdgcn_cvt_pkrtz(a, b);
    hi = __builtin_bit_cast(uint32_t, h);
    auto l = __builtin_amdgcn_cvt_pkrtz(a - (float)h[0], b - (float)h[1]);
    lo = __builtin_bit_cast(uint32_t, l);
}
#define ALM_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0)

// Row operand X [R][L] (f32) -> per-row power-of-two scale, f16 hi / lo images in A-fragment order:
// granule ((s * 2 + h) * Rpad + r) holds k = 16 s + 8 h + 0..7 of row r (zero beyond L / R).
// `neg_odd`: negate odd k (the imaginary parts of alm: out = are Yre - aim Yim).
// Two kernels so that long rows (gout: Npix values) spread over the chip: row_absmax_kernel
// (grid = rows x segments, integer atomicMax on the bit patterns of |x|: order-independent) and
// split_rows_kernel (same grid).
constexpr int SPLIT_SEG = 4096;                        // values per block

__global__ void __launch_bounds__(256)
row_absmax_kernel(const float* __restrict__ X, int R, int L, unsigned int* __restrict__ rowmax)
{
    const int r = blockIdx.x, tid = threadIdx.x;
    const int k0 = blockIdx.y * SPLIT_SEG, k1 = min(L, k0 + SPLIT_SEG);
    float m = 0.f;
    for (int i = k0 + tid; i < k1; i += 256) m = fmaxf(m, fabsf(X[(size_t)r * L + i]));
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((tid & 63) == 0 && m > 0.f) atomicMax(rowmax + r, __float_as_uint(m));
}

__global__ void __launch_bounds__(256)
split_rows_kernel(const float* __restrict__ X, int R, int L, int Rpad, int neg_odd,
                  const unsigned int* __restrict__ rowmax, uint4* __restrict__ img_hi,
                  uint4* __restrict__ img_lo, float* __restrict__ inv_scale)
{
    const int r = blockIdx.x, tid = threadIdx.x;
    const int nsteps = (L + 15) / 16;
    float scale = 1.0f;
    if (r < R) {
        const float m = __uint_as_float(rowmax[r]);
        if (m > 0.f) scale = exp2f(fminf(fmaxf(floorf(log2f(8192.0f / m)), -100.f), 100.f));   // max|x| * scale in [2^12, 2^13]
    }
    if (blockIdx.y == 0 && tid == 0) inv_scale[r] = (r < R) ? 1.0f / scale : 0.f;
    const int g0 = blockIdx.y * (SPLIT_SEG / 8), g1 = min(nsteps * 2, g0 + SPLIT_SEG / 8);
    for (int g = g0 + tid; g < g1; g += 256) {
        float v[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int k = g * 8 + jj;
            float x = (r < R && k < L) ? X[(size_t)r * L + k] * scale : 0.f;
            v[jj] = (neg_odd && (jj & 1)) ? -x : x;
        }
        uint4 hi, lo;
        split2h(v[0], v[1], hi.x, lo.x); split2h(v[2], v[3], hi.y, lo.y);
        split2h(v[4], v[5], hi.z, lo.z); split2h(v[6], v[7], hi.w, lo.w);
        img_hi[(size_t)g * Rpad + r] = hi;
        img_lo[(size_t)g * Rpad + r] = lo;
    }
}

// Long rows (round 3; gout: 128 rows of 49 152 values in front of every C3-shape backward): the two kernels above cost 64 us
// there, mostly because every 16-byte granule store of split_rows_kernel lands 2 KB from the next (granule-major images,
// one block per row).  Two launches without atomics or memset instead: the scale of every row by one block per row
// (coalesced float4 reads), then a TILED split -- a block reads 64 rows x 32 granules along the rows (a wave reads 2 KB
// contiguous pieces), turns the tile through LDS and writes it granule-major, 64 rows x 16 B = 1 KB contiguous per store.
__global__ void __launch_bounds__(256)
row_scale_alm_kernel(const float* __restrict__ X, int R, int L, float* __restrict__ scale, float* __restrict__ inv_scale)
{
    __shared__ float wmax[4];
    const int r = blockIdx.x, tid = threadIdx.x;
    float m = 0.f;
    if (r < R) {
        const float* row = X + (size_t)r * L;
        if ((L & 3) == 0 && (reinterpret_cast<uintptr_t>(row) & 15) == 0) {
            for (int i = tid; i < L / 4; i += 256) {
                const float4 v = reinterpret_cast<const float4*>(row)[i];
                m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
            }
        } else {
            for (int i = tid; i < L; i += 256) m = fmaxf(m, fabsf(row[i]));
        }
    }
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((tid & 63) == 0) wmax[tid >> 6] = m;
    __syncthreads();
    if (tid == 0) {
        m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
        float sc = 1.0f;
        if (r < R && m > 0.f) sc = exp2f(fminf(fmaxf(floorf(log2f(8192.0f / m)), -100.f), 100.f));   // max|x| * scale in [2^12, 2^13]
        scale[r] = sc;
        inv_scale[r] = (r < R) ? 1.0f / sc : 0.f;
    }
}

__global__ void __launch_bounds__(256)
split_rows_tiled_kernel(const float* __restrict__ X, int R, int L, int Rpad, int neg_odd, const float* __restrict__ scale,
                        uint4* __restrict__ img_hi, uint4* __restrict__ img_lo)
{
    __shared__ uint4 t_hi[32 * 65], t_lo[32 * 65];     // [granule][row], rows padded to 65: the turn is (nearly) conflict-free
    const int tid = threadIdx.x;
    const int g0 = blockIdx.x * 32, r0 = blockIdx.y * 64;
    const int ngran = ((L + 15) / 16) * 2;
    const int gl = tid & 31;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int rr = (tid >> 5) + 8 * it, r = r0 + rr, g = g0 + gl;
        float v[8];
        const float sc = r < R ? scale[r] : 0.f;
        const size_t base = (size_t)r * L + (size_t)g * 8;
        if (r < R && g * 8 + 7 < L && ((base & 3) == 0)) {
            const float4 a = *reinterpret_cast<const float4*>(X + base), b = *reinterpret_cast<const float4*>(X + base + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        } else {
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) v[jj] = (r < R && g * 8 + jj < L) ? X[base + jj] : 0.f;
        }
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) { v[jj] *= sc; if (neg_odd && (jj & 1)) v[jj] = -v[jj]; }
        uint4 hi, lo;
        split2h(v[0], v[1], hi.x, lo.x); split2h(v[2], v[3], hi.y, lo.y);
        split2h(v[4], v[5], hi.z, lo.z); split2h(v[6], v[7], hi.w, lo.w);
        t_hi[gl * 65 + rr] = hi;
        t_lo[gl * 65 + rr] = lo;
    }
    __syncthreads();
    const int rr = tid & 63;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int gq = (tid >> 6) + 4 * it, g = g0 + gq;
        if (g < ngran) {
            img_hi[(size_t)g * Rpad + r0 + rr] = t_hi[gq * 65 + rr];
            img_lo[(size_t)g * Rpad + r0 + rr] = t_lo[gq * 65 + rr];
        }
    }
}

// forward: out[r, j] = sum_k A[r, k] Y[k, j], k = (c, re|im).  Block = 4 waves x 32 pixels, MT row
// tiles per wave; per chunk of 32 coefficients (4 K steps) the block copies the pre-split A
// granules to LDS (fetched one chunk ahead into registers) and every lane splits the 16 Ylm values
// it loaded (also one chunk ahead): 12 MT MFMAs against ~130 VALU instructions per chunk and wave.
template <int MT>
__global__ void __launch_bounds__(256)
alm2pix_fwd_f16_kernel(const uint4* __restrict__ a_hi, const uint4* __restrict__ a_lo,
                       const float* __restrict__ inv_scale, const float* __restrict__ Ylm, float y_scale,
                       int R, int Rpad, int Ncoeff, int Npix, float* __restrict__ out)
{
    constexpr int ROWS = MT * 32;
    constexpr int NG = 8 * ROWS;                       // granules per image and chunk: (ks, h, row)
    constexpr int PT = NG / 256;                       // per thread
    __shared__ uint4 lds_hi[NG], lds_lo[NG];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r0 = blockIdx.y * ROWS;
    const int j0 = (blockIdx.x * 4 + wave) * 32;
    const int jl = min(j0 + (lane & 31), Npix - 1);
    const int h = lane >> 5;
    const int nsteps = (2 * Ncoeff + 15) / 16;
    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[m][e] = 0.f;

    float2 bcur[16], bnxt[16];                         // [ks * 4 + i]: coefficient c0 + 8 ks + 4 h + i
    uint4 ahq[PT], alq[PT];
    auto load_b = [&](int c0, float2 (&b)[16]) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int c = c0 + 8 * (u >> 2) + 4 * h + (u & 3);
            b[u] = (c < Ncoeff) ? *reinterpret_cast<const float2*>(Ylm + ((size_t)c * Npix + jl) * 2)
                                : make_float2(0.f, 0.f);
        }
    };
    auto fetch_a = [&](int s0) {                       // K steps s0 .. s0 + 3
#pragma unroll
        for (int u = 0; u < PT; ++u) {
            const int i = tid + u * 256;
            const int row = i % ROWS, g = i / ROWS;    // g = ks * 2 + h
            const int sg = s0 * 2 + g;
            const bool ok = sg < nsteps * 2;
            ahq[u] = ok ? a_hi[(size_t)sg * Rpad + r0 + row] : make_uint4(0, 0, 0, 0);
            alq[u] = ok ? a_lo[(size_t)sg * Rpad + r0 + row] : make_uint4(0, 0, 0, 0);
        }
    };
    load_b(0, bcur);
    fetch_a(0);
    for (int c0 = 0; c0 < Ncoeff; c0 += 32) {
#if !defined(RIME_ALM_ABL) || RIME_ALM_ABL != 2
        __syncthreads();                               // previous chunk consumed
#pragma unroll
        for (int u = 0; u < PT; ++u) { lds_hi[tid + u * 256] = ahq[u]; lds_lo[tid + u * 256] = alq[u]; }
        __syncthreads();
        if (c0 + 32 < Ncoeff) { load_b(c0 + 32, bnxt); fetch_a(c0 / 8 + 4); }
#else
        if (c0 + 32 < Ncoeff) { load_b(c0 + 32, bnxt); }
#endif
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            uint4 bh, bl;
#if defined(RIME_ALM_ABL) && RIME_ALM_ABL == 1
            bh = make_uint4(__float_as_uint(bcur[ks * 4 + 0].x), __float_as_uint(bcur[ks * 4 + 1].x), __float_as_uint(bcur[ks * 4 + 2].x), __float_as_uint(bcur[ks * 4 + 3].x));
            bl = make_uint4(__float_as_uint(bcur[ks * 4 + 0].y), __float_as_uint(bcur[ks * 4 + 1].y), __float_as_uint(bcur[ks * 4 + 2].y), __float_as_uint(bcur[ks * 4 + 3].y));
#else
            split2h(bcur[ks * 4 + 0].x * y_scale, bcur[ks * 4 + 0].y * y_scale, bh.x, bl.x);
            split2h(bcur[ks * 4 + 1].x * y_scale, bcur[ks * 4 + 1].y * y_scale, bh.y, bl.y);
            split2h(bcur[ks * 4 + 2].x * y_scale, bcur[ks * 4 + 2].y * y_scale, bh.z, bl.z);
            split2h(bcur[ks * 4 + 3].x * y_scale, bcur[ks * 4 + 3].y * y_scale, bh.w, bl.w);
#endif
#pragma unroll
            for (int m = 0; m < MT; ++m) {
#if defined(RIME_ALM_ABL) && RIME_ALM_ABL == 2
                const uint4 ah = make_uint4(m, ks, h, 1), al = make_uint4(ks, m, 2, h);
#else
                const int gi = (ks * 2 + h) * ROWS + m * 32 + (lane & 31);
                const uint4 ah = lds_hi[gi], al = lds_lo[gi];
#endif
#if defined(RIME_ALM_ABL) && RIME_ALM_ABL == 3
                acc[m][0] += __uint_as_float(ah.x ^ bh.x ^ al.y ^ bl.z ^ bh.w ^ bl.x ^ bh.y ^ bh.z ^ bl.y ^ bl.w);
#else
                acc[m] = ALM_MFMA(ah, bh, acc[m]);
                acc[m] = ALM_MFMA(ah, bl, acc[m]);
                acc[m] = ALM_MFMA(al, bh, acc[m]);
#endif
            }
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) bcur[u] = bnxt[u];
    }
    RIME_MFMA_SETTLE();
    const int col = j0 + (lane & 31);
    if (col < Npix) {
        const float iy = 1.0f / y_scale;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = r0 + m * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (row < R) out[(size_t)row * Npix + col] = acc[m][e] * inv_scale[row] * iy;
            }
    }
}

// backward: galm[r, c, q] = sum_j gout[r, j] Y[c, j, q] (q ? -1 : +1).  K = pixels (16 per MFMA).
// Wave = 32 coefficients x MT row tiles with separate accumulators for the re and im columns (a
// B fragment needs 8 consecutive pixels of ONE coefficient row, and both components are used).
// Block = 4 waves = 128 coefficients.  Per chunk of 32 pixels the block stages the Ylm tile
// [128 coefficients][32 px][re,im] through LDS -- global loads are 256-B row segments, 16 B per lane,
// fetched one chunk ahead -- together with the pre-split gout granules; the 32-pixel chunks are
// dealt cyclically to S blocks (partials reduced by alm_reduce_kernel).  Measured 2.0 TB/s of Ylm
// at R = 128 and 3.5 TB/s at R = 4: latency-bound (one chunk = 32 KB per block in flight, ~7 us
// loaded latency); per-lane row streaming without LDS staging and contiguous split ranges are
// within 5 % of this.
template <int MT>
__global__ void __launch_bounds__(256, 2)       // two blocks per CU: the kernel is bound by the Ylm bytes in flight
alm2pix_bwd_f16_kernel(const uint4* __restrict__ g_hi, const uint4* __restrict__ g_lo,
                       const float* __restrict__ inv_scale, const float* __restrict__ Ylm, float y_scale,
                       int R, int Rpad, int Ncoeff, int Npix, int S, float* __restrict__ part)
{
    constexpr int ROWS = MT * 32;
    constexpr int NG = 4 * ROWS;                       // gout granules per image and chunk of 2 K steps
    constexpr int PT = (NG + 255) / 256;
    constexpr int YROW = 272;                          // 32 px x 8 B + 16 B pad: conflict-free 16-B reads
    __shared__ uint4 lds_hi[NG], lds_lo[NG];
    __shared__ __align__(16) unsigned char y_lds[128 * YROW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // 32-pixel chunks are dealt cyclically to the S blocks of a coefficient tile, and the split index is
    // the fastest grid dimension: blocks that run together read ADJACENT 256-B segments of the same
    // Ylm rows (contiguous split ranges stream 128 x S scattered segments: 2.2 instead of 3+ TB/s)
    const int split = blockIdx.x;
    const int r0 = blockIdx.z * ROWS;
    const int cblk = blockIdx.y * 128;
    const int c = cblk + wave * 32 + (lane & 31);
    const int h = lane >> 5;
    const int nsteps = (Npix + 15) / 16;
    const int send = nsteps;
    f32x16 accr[MT], acci[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) { accr[m][e] = 0.f; acci[m][e] = 0.f; }

    float4 yq[8];                                      // this thread's 8 x 16 B of the next Ylm tile
    uint4 ahq[PT], alq[PT];
    const bool aligned = (Npix & 1) == 0;              // rows start 16-B aligned
    auto fetch_y = [&](int s0) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = tid + u * 256;               // (row, 16-B segment): 2 pixels
            const int row = i >> 4, seg = i & 15;
            const int cc = cblk + row, j = 16 * s0 + 2 * seg;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (cc < Ncoeff && j < Npix) {
                const float* src = Ylm + ((size_t)cc * Npix + j) * 2;
                if (j + 1 < Npix) {
                    if (aligned) v = *reinterpret_cast<const float4*>(src);
                    else v = make_float4(src[0], src[1], src[2], src[3]);
                } else { v.x = src[0]; v.y = src[1]; }
            }
            yq[u] = v;
        }
    };
    auto fetch_a = [&](int s0) {
#pragma unroll
        for (int u = 0; u < PT; ++u) {
            const int i = tid + u * 256;
            const int row = i % ROWS, g = i / ROWS;    // g = ks * 2 + h
            const int sg = s0 * 2 + g;
            const bool ok = i < NG && sg < send * 2;
            ahq[u] = ok ? g_hi[(size_t)sg * Rpad + r0 + row] : make_uint4(0, 0, 0, 0);
            alq[u] = ok ? g_lo[(size_t)sg * Rpad + r0 + row] : make_uint4(0, 0, 0, 0);
        }
    };
    if (2 * split < send) { fetch_y(2 * split); fetch_a(2 * split); }
    const unsigned char* ymine = y_lds + (wave * 32 + (lane & 31)) * YROW + 64 * h;
    for (int s0 = 2 * split; s0 < send; s0 += 2 * S) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < PT; ++u)
            if (tid + u * 256 < NG) { lds_hi[tid + u * 256] = ahq[u]; lds_lo[tid + u * 256] = alq[u]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = tid + u * 256;
            *reinterpret_cast<float4*>(y_lds + (i >> 4) * YROW + (i & 15) * 16) = yq[u];
        }
        __syncthreads();
        if (s0 + 2 * S < send) { fetch_y(s0 + 2 * S); fetch_a(s0 + 2 * S); }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            float4 y[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) y[q] = *reinterpret_cast<const float4*>(ymine + 128 * ks + 16 * q);
            uint4 rh, rl, ih, il;
            split2h(y[0].x * y_scale, y[0].z * y_scale, rh.x, rl.x);
            split2h(y[1].x * y_scale, y[1].z * y_scale, rh.y, rl.y);
            split2h(y[2].x * y_scale, y[2].z * y_scale, rh.z, rl.z);
            split2h(y[3].x * y_scale, y[3].z * y_scale, rh.w, rl.w);
            split2h(y[0].y * y_scale, y[0].w * y_scale, ih.x, il.x);
            split2h(y[1].y * y_scale, y[1].w * y_scale, ih.y, il.y);
            split2h(y[2].y * y_scale, y[2].w * y_scale, ih.z, il.z);
            split2h(y[3].y * y_scale, y[3].w * y_scale, ih.w, il.w);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int gi = (ks * 2 + h) * ROWS + m * 32 + (lane & 31);
                const uint4 ah = lds_hi[gi], al = lds_lo[gi];
                accr[m] = ALM_MFMA(ah, rh, accr[m]);
                acci[m] = ALM_MFMA(ah, ih, acci[m]);
                accr[m] = ALM_MFMA(ah, rl, accr[m]);
                acci[m] = ALM_MFMA(ah, il, acci[m]);
                accr[m] = ALM_MFMA(al, rh, accr[m]);
                acci[m] = ALM_MFMA(al, ih, acci[m]);
            }
        }
    }
    RIME_MFMA_SETTLE();
    if (c < Ncoeff) {
        float* dst = part + (size_t)split * R * Ncoeff * 2;
        const float iy = 1.0f / y_scale;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = r0 + m * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (row < R) {
                    const float sc = inv_scale[row] * iy;
                    *reinterpret_cast<float2*>(dst + ((size_t)row * Ncoeff + c) * 2) =
                        make_float2(accr[m][e] * sc, -acci[m][e] * sc);
                }
            }
    }
}

// backward, LDS-DMA form (even Npix): the kernel above is bound by the Ylm bytes it keeps in flight (one 32-KB tile
// per block in registers, two blocks per CU: 64 KB -> 2.8 TB/s).  Here the Ylm tile and the pre-split gout granules of a
// chunk (KS K steps of 16 pixels) go from global memory straight into LDS (global_load_lds_dwordx4: no registers hold
// bytes in flight) through a ring of NSLOT slots, ONE raw barrier per chunk and a counted s_waitcnt vmcnt that leaves
// the NSLOT - 2 younger chunks in flight across it (MI355X guide, "Pipelining across barriers"), one 4-wave block per CU.
// An LDS-DMA wave instruction deposits 64 x 16 B contiguously in lane order, so rows cannot be padded; the fragment
// reads stay conflict-free through an XOR swizzle applied to the SOURCE address: the chunk's tile is 128 rows of
// GR = 8 KS granules (two pixels each), granule s of row c lands at position s ^ (c & 15) (KS 2) or s ^ ((c >> 1) & 7)
// (KS 1: two rows per 256 B of banks).  Rows / pixels beyond the
// matrix read a 16-byte block of zeros.  Every wave issues the same number of DMA instructions per chunk (the counted
// wait depends on it): a remainder is padded with loads of the zero block into a scratch KB.
// The load is issued from inline asm: hipcc counts a builtin LDS-DMA as a pending LDS write and drains it (vmcnt(0))
// before the next ds_read, which would serialise the ring; the kernel counts its own DMA queue.
__device__ __forceinline__ void glds16(const void* src, unsigned lds_wave_base)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(lds_wave_base) : "memory");
}

template <int N> __device__ __forceinline__ void wait_vmcnt()
{
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// wait until all but the `younger` most recent chunks (NPW instructions each) of this wave have landed
template <int NPW, int MAXY> __device__ __forceinline__ void wait_chunks(int younger)
{
    if constexpr (MAXY == 0) wait_vmcnt<0>();
    else {
        if (younger >= MAXY) wait_vmcnt<NPW * MAXY>();
        else wait_chunks<NPW, MAXY - 1>(younger);
    }
}

template <int MT, int KS, int NSLOT> struct BwdDma {
    static constexpr int ROWS = MT * 32;
    static constexpr int GR = 8 * KS;                  // 16-byte granules (two pixels) per row of the chunk's Ylm tile
    static constexpr int RP = 64 / GR;                 // tile rows per wave instruction (1 KB)
    static constexpr int YB = 128 * GR * 16;           // the chunk's Ylm tile: 128 coefficients x KS x 16 px x (re, im) f32
    static constexpr int GB = 2 * ROWS * 16;           // one K step of one gout image (hi or lo)
    static constexpr int SLOT = YB + KS * 2 * GB;
    static constexpr int NGW = GB / 1024;              // wave instructions per gout image and K step
    static constexpr int TI = KS * (16 + 2 * NGW);     // wave instructions per chunk
    static constexpr int NPW = (TI + 3) / 4;           // per wave
    static constexpr int LDS = NSLOT * SLOT + 1024;    // + the scratch KB of the padding loads
    static_assert(LDS <= 160 * 1024, "LDS ring too large");
    static_assert(NPW * (NSLOT - 2) < 64, "vmcnt range");
};

template <int MT, int KS, int NSLOT>
__global__ void __launch_bounds__(256, 1)
alm2pix_bwd_f16_dma_kernel(const uint4* __restrict__ g_hi, const uint4* __restrict__ g_lo,
                           const float* __restrict__ inv_scale, const float* __restrict__ Ylm,
                           const float* __restrict__ zero16, float y_scale,
                           int R, int Rpad, int Ncoeff, int Npix, int S, int CT, int RT, float* __restrict__ part)
{
    using C = BwdDma<MT, KS, NSLOT>;
    constexpr int ROWS = C::ROWS, AHEAD = NSLOT - 1;
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned smem_addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    // Workgroups go to the 8 XCDs round-robin by linear id.  The blocks of one pixel split read the SAME gout chunks
    // (and, over row tiles, the same Ylm tile): each XCD takes a contiguous range of the (split, coefficient tile,
    // row tile) order, so the ~32 blocks an XCD holds at a time share a split and the gout stream is served by that
    // XCD's L2 instead of crossing the fabric once per coefficient tile (it was half as many bytes as Ylm itself).
    const int NB = S * CT * RT, per_xcd = (NB + 7) / 8;
    const int q = (int)(blockIdx.x % 8) * per_xcd + (int)(blockIdx.x / 8);
    if ((int)(blockIdx.x / 8) >= per_xcd || q >= NB) return;      // whole block, before any barrier
    const int split = q / (CT * RT);
    const int r0 = (q % RT) * ROWS;
    const int cblk = ((q / RT) % CT) * 128;
    const int cl = wave * 32 + (lane & 31);            // this lane's coefficient row of the tile
    const int c = cblk + cl;
    const int h = lane >> 5;
    const int send = (Npix + 15) / 16;                 // K steps of 16 pixels
    const int nchunk = (send + KS - 1) / KS;
    const int niter = split < nchunk ? (nchunk - split + S - 1) / S : 0;      // chunks split, split + S, ...
    f32x16 accr[MT], acci[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) { accr[m][e] = 0.f; acci[m][e] = 0.f; }

    auto issue = [&](int it, int slot) {
        const int s0 = KS * (split + it * S);
        const unsigned base = smem_addr + (unsigned)(slot * C::SLOT);
#pragma unroll
        for (int u = 0; u < C::NPW; ++u) {
            const int t = wave + 4 * u;                // wave-uniform
            if (t < 16 * KS) {                         // RP whole rows of the chunk's tile
                const int cr = C::RP * t + lane / C::GR;
                const int sg = (lane % C::GR) ^ ((cr >> (KS == 1 ? 1 : 0)) & (C::GR - 1));
                const int cc = cblk + cr, j = 16 * s0 + 2 * sg;
                const float* src = (cc < Ncoeff && j < Npix) ? Ylm + ((size_t)cc * Npix + j) * 2 : zero16;
                glds16(src, base + t * 1024);
            } else if (t < C::TI) {
                const int tg = t - 16 * KS;
                const int ks = tg / (2 * C::NGW), r = tg % (2 * C::NGW), img = r / C::NGW, k = r % C::NGW;
                const int i = k * 64 + lane;
                const int row = i % ROWS, hh = i / ROWS;
                const int qq = (s0 + ks) * 2 + hh;
                const uint4* g = img ? g_lo : g_hi;
                const void* src = s0 + ks < send ? (const void*)&g[(size_t)qq * Rpad + r0 + row] : (const void*)zero16;
                glds16(src, base + C::YB + (ks * 2 + img) * C::GB + k * 1024);
            } else {
                glds16(zero16, smem_addr + NSLOT * C::SLOT);
            }
        }
    };
#pragma unroll
    for (int a = 0; a < AHEAD; ++a)
        if (a < niter) issue(a, a);
    const uint32_t ybase = (uint32_t)cl * (C::GR * 16u);
    const uint32_t sw = (uint32_t)((cl >> (KS == 1 ? 1 : 0)) & (C::GR - 1));
    int slot = 0, nslot = AHEAD;                       // slot being read; slot the next issue fills
    for (int it = 0; it < niter; ++it) {
        wait_chunks<C::NPW, AHEAD - 1>(niter - 1 - it);          // chunk `it` of this wave has landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                  // every wave's part has landed; everyone is done with chunk it - 1
        if (it + AHEAD < niter) issue(it + AHEAD, nslot);        // into the slot chunk it - 1 occupied
        const unsigned char* sl = smem + slot * C::SLOT;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const uint4* lds_hi = reinterpret_cast<const uint4*>(sl + C::YB + ks * 2 * C::GB);
            const uint4* lds_lo = reinterpret_cast<const uint4*>(sl + C::YB + (ks * 2 + 1) * C::GB);
            float4 y[4];
#pragma unroll
            for (int qd = 0; qd < 4; ++qd)
                y[qd] = *reinterpret_cast<const float4*>(sl + ybase + ((((uint32_t)(8 * ks + 4 * h + qd)) ^ sw) * 16u));
            uint4 rh, rl, ih, il;
            split2h(y[0].x * y_scale, y[0].z * y_scale, rh.x, rl.x);
            split2h(y[1].x * y_scale, y[1].z * y_scale, rh.y, rl.y);
            split2h(y[2].x * y_scale, y[2].z * y_scale, rh.z, rl.z);
            split2h(y[3].x * y_scale, y[3].z * y_scale, rh.w, rl.w);
            split2h(y[0].y * y_scale, y[0].w * y_scale, ih.x, il.x);
            split2h(y[1].y * y_scale, y[1].w * y_scale, ih.y, il.y);
            split2h(y[2].y * y_scale, y[2].w * y_scale, ih.z, il.z);
            split2h(y[3].y * y_scale, y[3].w * y_scale, ih.w, il.w);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int gi = h * ROWS + m * 32 + (lane & 31);
                const uint4 ah = lds_hi[gi], al = lds_lo[gi];
                accr[m] = ALM_MFMA(ah, rh, accr[m]);
                acci[m] = ALM_MFMA(ah, ih, acci[m]);
                accr[m] = ALM_MFMA(ah, rl, accr[m]);
                acci[m] = ALM_MFMA(ah, il, acci[m]);
                accr[m] = ALM_MFMA(al, rh, accr[m]);
                acci[m] = ALM_MFMA(al, ih, acci[m]);
            }
        }
        slot = slot + 1 == NSLOT ? 0 : slot + 1;
        nslot = nslot + 1 == NSLOT ? 0 : nslot + 1;
    }
    RIME_MFMA_SETTLE();
    if (c < Ncoeff) {
        float* dst = part + (size_t)split * R * Ncoeff * 2;
        const float iy = 1.0f / y_scale;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = r0 + m * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (row < R) {
                    const float sc = inv_scale[row] * iy;
                    *reinterpret_cast<float2*>(dst + ((size_t)row * Ncoeff + c) * 2) =
                        make_float2(accr[m][e] * sc, -acci[m][e] * sc);
                }
            }
    }
}

// ---------------------------------------------------------------------------------------
// PACKED Ylm (round 3): Ylm x y_scale split ONCE into f16 hi / lo halves and stored in MFMA fragment order -- the
// same 8 bytes per (coefficient, pixel) as the complex64 matrix, one copy per direction (the forward contracts over
// coefficients, the backward over pixels: the 16-byte granule a lane feeds to the matrix core holds 8 consecutive K
// values, so the two directions need transposed granules).  The GEMM kernels then do no arithmetic on the streamed
// operand at all: a wave's fragment is ONE fully coalesced 1-KB load (forward: into registers; backward: LDS-DMA into
// the wave's own ring region, read back with linear, conflict-free ds_read_b128).  Callers cache the packed copies per
// Ylm object (bayeslim_amd/ops.py); a new or modified Ylm is packed again.
//   forward  Yf: [pixel tile jt of 32][K step s of 8 coefficients][hi | lo][lane][8 x f16]
//                lane (n = lane & 31 -> pixel 32 jt + n, h = lane >> 5): k = (c, q) = (8 s + 4 h + i, re | im), i = 0..3
//   backward Yb: [coefficient tile ct of 128][K step s of 16 pixels][wave w][re_hi, re_lo, im_hi, im_lo][lane][8 x f16]
//                lane (c = 128 ct + 32 w + (lane & 31), h = lane >> 5): pixels 16 s + 8 h + 0..7
//   rows / pixels / K steps beyond the matrix hold zeros; the K-step count is padded to a whole number of chunks
// ---------------------------------------------------------------------------------------
constexpr int PK_FWD_CHUNK = 4;                        // forward: K steps per chunk (32 coefficients)
constexpr int PK_BWD_KS = 2;                           // backward: K steps per ring chunk (32 pixels)

__host__ __device__ inline int pk_fwd_steps(int Ncoeff) { const int n = (2 * Ncoeff + 15) / 16; return (n + PK_FWD_CHUNK - 1) / PK_FWD_CHUNK * PK_FWD_CHUNK; }
__host__ __device__ inline int pk_bwd_steps(int Npix) { const int n = (Npix + 15) / 16; return (n + PK_BWD_KS - 1) / PK_BWD_KS * PK_BWD_KS; }

__global__ void __launch_bounds__(256)
alm_pack_fwd_kernel(const float* __restrict__ Ylm, float y_scale, int Ncoeff, int Npix, int nsteps, size_t ntile,
                    uint4* __restrict__ Yf)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;          // (jt, s, lane)
    const int lane = (int)(i & 63);
    const size_t js = i >> 6;
    const int s = (int)(js % nsteps);
    const size_t jt = js / nsteps;
    if (jt >= ntile) return;                           // (tiles beyond the map are written too: zeros)
    const size_t j = jt * 32 + (lane & 31);
    const int h = lane >> 5;
    float v[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = 8 * s + 4 * h + q;
        float2 y = make_float2(0.f, 0.f);
        if (c < Ncoeff && j < (size_t)Npix) y = *reinterpret_cast<const float2*>(Ylm + ((size_t)c * Npix + j) * 2);
        v[2 * q] = y.x * y_scale; v[2 * q + 1] = y.y * y_scale;
    }
    uint4 hi, lo;
    split2h(v[0], v[1], hi.x, lo.x); split2h(v[2], v[3], hi.y, lo.y);
    split2h(v[4], v[5], hi.z, lo.z); split2h(v[6], v[7], hi.w, lo.w);
    uint4* dst = Yf + ((jt * nsteps + s) * 2) * 64 + lane;
    dst[0] = hi;
    dst[64] = lo;
}

__global__ void __launch_bounds__(256)
alm_pack_bwd_kernel(const float* __restrict__ Ylm, float y_scale, int Ncoeff, int Npix, int nsteps, uint4* __restrict__ Yb)
{
    // block = (ct, s): thread = (wave w, lane); reads 8 pixels x (re, im) = 64 contiguous bytes of one coefficient row
    const int s = (int)(blockIdx.x % nsteps), ct = (int)(blockIdx.x / nsteps);
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = ct * 128 + w * 32 + (lane & 31);
    const int j0 = 16 * s + 8 * (lane >> 5);
    float re[8], im[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        float2 y = make_float2(0.f, 0.f);
        if (c < Ncoeff && j0 + q < Npix) y = *reinterpret_cast<const float2*>(Ylm + ((size_t)c * Npix + j0 + q) * 2);
        re[q] = y.x * y_scale; im[q] = y.y * y_scale;
    }
    uint4 rh, rl, ih, il;
    split2h(re[0], re[1], rh.x, rl.x); split2h(re[2], re[3], rh.y, rl.y);
    split2h(re[4], re[5], rh.z, rl.z); split2h(re[6], re[7], rh.w, rl.w);
    split2h(im[0], im[1], ih.x, il.x); split2h(im[2], im[3], ih.y, il.y);
    split2h(im[4], im[5], ih.z, il.z); split2h(im[6], im[7], ih.w, il.w);
    uint4* dst = Yb + ((((size_t)ct * nsteps + s) * 4 + w) * 4) * 64 + lane;
    dst[0] = rh; dst[64] = rl; dst[128] = ih; dst[192] = il;
}

// forward on the packed copy: block = 4 waves x 32 pixels, MT row tiles per wave.  A (the pre-split alm granules) goes
// through LDS, double buffered (one barrier per chunk of 4 K steps); the B fragments are coalesced 16-byte loads per
// lane from the wave's own contiguous stream: the two registers of a K step are reloaded, right after their MFMAs, with
// the same K step of the chunk TWO ahead, so a wave keeps 4 to 8 K steps (8 - 16 KB) of its stream in flight.
template <int MT>
__global__ void __launch_bounds__(256)
alm2pix_fwd_packed_kernel(const uint4* __restrict__ a_hi, const uint4* __restrict__ a_lo,
                          const float* __restrict__ inv_scale, const uint4* __restrict__ Yf, float y_scale,
                          int R, int Rpad, int Ncoeff, int Npix, int nsteps, int chunks_per_split, float* __restrict__ out)
{
    // blockIdx.z = K split: chunks [z * chunks_per_split, ...) of the coefficient axis; the split's partial result goes
    // to plane z of `out` (S > 1: a workspace, summed in a fixed order by alm_reduce_kernel).  A wave owns a pixel tile, so
    // a map of Npix pixels offers only Npix / 32 waves: small maps are split over K as well (49 152 px: 384 blocks for
    // 512 resident ones -> 2 splits, - 6 %)
    constexpr int ROWS = MT * 32;
    constexpr int NG = 8 * ROWS;                       // A granules per image and chunk: (ks, h, row)
    constexpr int PT = NG / 256;
    __shared__ uint4 lds_hi[2][NG], lds_lo[2][NG];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r0 = blockIdx.y * ROWS;
    const size_t jt = (size_t)blockIdx.x * 4 + wave;
    const int j0 = (int)(jt * 32);
    const int asteps = (2 * Ncoeff + 15) / 16;         // K steps the A images hold
    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[m][e] = 0.f;
    const uint4* ysrc = Yf + (jt * nsteps * 2) * 64 + lane;            // this wave's stream: 2 KB per K step
    uint4 b0[8], b1[8];                                // even / odd chunk: [ks * 2 + (hi | lo)]
    uint4 ahq[PT], alq[PT];
    auto load_b = [&](int s0, uint4 (&b)[8]) {
#pragma unroll
        for (int u = 0; u < 8; ++u) b[u] = ysrc[(size_t)(s0 * 2 + u) * 64];
    };
    auto fetch_a = [&](int s0) {
#pragma unroll
        for (int u = 0; u < PT; ++u) {
            const int i = tid + u * 256;
            const int row = i % ROWS, g = i / ROWS;
            const int sg = s0 * 2 + g;
            const bool ok = sg < asteps * 2;
            ahq[u] = ok ? a_hi[(size_t)sg * Rpad + r0 + row] : make_uint4(0, 0, 0, 0);
            alq[u] = ok ? a_lo[(size_t)sg * Rpad + r0 + row] : make_uint4(0, 0, 0, 0);
        }
    };
    auto compute = [&](uint4 (&b)[8], int buf, int s_next) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int gi = (ks * 2 + (lane >> 5)) * ROWS + m * 32 + (lane & 31);
                const uint4 ah = lds_hi[buf][gi], al = lds_lo[buf][gi];
                acc[m] = ALM_MFMA(ah, b[2 * ks], acc[m]);
                acc[m] = ALM_MFMA(ah, b[2 * ks + 1], acc[m]);
                acc[m] = ALM_MFMA(al, b[2 * ks], acc[m]);
            }
            if (s_next >= 0) {                         // uniform
                b[2 * ks] = ysrc[(size_t)((s_next + ks) * 2) * 64];
                b[2 * ks + 1] = ysrc[(size_t)((s_next + ks) * 2 + 1) * 64];
            }
        }
    };
    const int ch0 = blockIdx.z * chunks_per_split;
    const int nchunk = min(nsteps / PK_FWD_CHUNK, ch0 + chunks_per_split);
    if (ch0 < nchunk) load_b(4 * ch0, b0);
    if (ch0 + 1 < nchunk) load_b(4 * (ch0 + 1), b1);
    fetch_a(4 * ch0);
    for (int ch = ch0; ch < nchunk; ch += 2) {
#pragma unroll
        for (int u = 0; u < PT; ++u) { lds_hi[0][tid + u * 256] = ahq[u]; lds_lo[0][tid + u * 256] = alq[u]; }
        __syncthreads();
        if (ch + 1 < nchunk) fetch_a(4 * (ch + 1));
        compute(b0, 0, ch + 2 < nchunk ? 4 * (ch + 2) : -1);
        if (ch + 1 < nchunk) {
#pragma unroll
            for (int u = 0; u < PT; ++u) { lds_hi[1][tid + u * 256] = ahq[u]; lds_lo[1][tid + u * 256] = alq[u]; }
            __syncthreads();
            if (ch + 2 < nchunk) fetch_a(4 * (ch + 2));
            compute(b1, 1, ch + 3 < nchunk ? 4 * (ch + 3) : -1);
        }
    }
    RIME_MFMA_SETTLE();
    const int col = j0 + (lane & 31);
    if (col < Npix) {
        const float iy = 1.0f / y_scale;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = r0 + m * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (row < R) out[((size_t)blockIdx.z * R + row) * Npix + col] = acc[m][e] * inv_scale[row] * iy;
            }
    }
}

// K splits of the packed forward (two 4-wave blocks per CU are resident), each split >= 16 chunks
static int fwd_packed_splits(int R, int Ncoeff, int Npix, int MT)
{
    static long resident = 0;
    if (resident == 0) {
        int dev = 0; hipDeviceProp_t prop;
        resident = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? 2L * prop.multiProcessorCount : 512;
        (void)hipGetLastError();
    }
    const long blocks = (long)((Npix + 127) / 128) * ((R + MT * 32 - 1) / (MT * 32));
    const int nchunk = pk_fwd_steps(Ncoeff) / PK_FWD_CHUNK;
    // measured at the C3 shape (384 blocks of 263 chunks, 512 resident): S 1 / 2 / 3 / 4 / 6 -> 0.911 / 0.853 / 0.890 /
    // 0.871 / 0.898 ms including the reduction of the partial planes: enough blocks for 1.5 resident grids, no more
    int best = (int)std::min<long>(4, std::max<long>(1, (3 * resident / 2 + blocks - 1) / blocks));
    while (best > 1 && nchunk / best < 16) --best;
    if (const char* e = getenv("RIME_ALM_FWD_SPLITS")) { const int v = atoi(e); if (v >= 1 && v <= 8 && nchunk / v >= 1) best = v; }   // lab
    return best;
}

// backward on the packed copy: the LDS-DMA ring of alm2pix_bwd_f16_dma_kernel with the chunk's Ylm tile replaced by
// the wave-private packed fragments (16 wave instructions of 1 KB per K step, contiguous in memory AND in the slot: the
// chunk is one 32-KB run) -- no swizzle, no split, no multiply; the gout granules are staged as before (shared by the
// four waves, hence the one barrier per chunk).
template <int MT, int NSLOT>
__global__ void __launch_bounds__(256, 1)
alm2pix_bwd_packed_kernel(const uint4* __restrict__ g_hi, const uint4* __restrict__ g_lo,
                          const float* __restrict__ inv_scale, const uint4* __restrict__ Yb,
                          const float* __restrict__ zero16, float y_scale,
                          int R, int Rpad, int Ncoeff, int Npix, int nsteps, int S, int CT, int RT, float* __restrict__ part)
{
    constexpr int KS = PK_BWD_KS;
    using C = BwdDma<MT, KS, NSLOT>;
    constexpr int ROWS = C::ROWS, AHEAD = NSLOT - 1;
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned smem_addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int NB = S * CT * RT, per_xcd = (NB + 7) / 8;
    const int q = (int)(blockIdx.x % 8) * per_xcd + (int)(blockIdx.x / 8);
    if ((int)(blockIdx.x / 8) >= per_xcd || q >= NB) return;      // whole block, before any barrier
    const int split = q / (CT * RT);
    const int r0 = (q % RT) * ROWS;
    const int ct = (q / RT) % CT;
    const int c = ct * 128 + wave * 32 + (lane & 31);
    const int h = lane >> 5;
    const int send = (Npix + 15) / 16;                 // K steps the gout images hold
    const int nchunk = nsteps / KS;
    const int niter = split < nchunk ? (nchunk - split + S - 1) / S : 0;
    f32x16 accr[MT], acci[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) { accr[m][e] = 0.f; acci[m][e] = 0.f; }

    const uint4* ytile = Yb + (size_t)ct * nsteps * 16 * 64 + lane;
    auto issue = [&](int it, int slot) {
        const int s0 = KS * (split + it * S);
        const unsigned base = smem_addr + (unsigned)(slot * C::SLOT);
#pragma unroll
        for (int u = 0; u < C::NPW; ++u) {
            const int t = wave + 4 * u;                // wave-uniform
            if (t < 16 * KS) {                         // KB number t of the chunk's 32-KB run: (ks, w, plane) = t
                glds16(ytile + ((size_t)s0 * 16 + t) * 64, base + t * 1024);
            } else if (t < C::TI) {
                const int tg = t - 16 * KS;
                const int ks = tg / (2 * C::NGW), r = tg % (2 * C::NGW), img = r / C::NGW, k = r % C::NGW;
                const int i = k * 64 + lane;
                const int row = i % ROWS, hh = i / ROWS;
                const int qq = (s0 + ks) * 2 + hh;
                const uint4* g = img ? g_lo : g_hi;
                const void* src = s0 + ks < send ? (const void*)&g[(size_t)qq * Rpad + r0 + row] : (const void*)zero16;
                glds16(src, base + C::YB + (ks * 2 + img) * C::GB + k * 1024);
            } else {
                glds16(zero16, smem_addr + NSLOT * C::SLOT);
            }
        }
    };
#pragma unroll
    for (int a = 0; a < AHEAD; ++a)
        if (a < niter) issue(a, a);
    int slot = 0, nslot = AHEAD;
    for (int it = 0; it < niter; ++it) {
        wait_chunks<C::NPW, AHEAD - 1>(niter - 1 - it);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (it + AHEAD < niter) issue(it + AHEAD, nslot);
        const unsigned char* sl = smem + slot * C::SLOT;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const uint4* lds_hi = reinterpret_cast<const uint4*>(sl + C::YB + ks * 2 * C::GB);
            const uint4* lds_lo = reinterpret_cast<const uint4*>(sl + C::YB + (ks * 2 + 1) * C::GB);
            const uint4* yf = reinterpret_cast<const uint4*>(sl + (ks * 16 + wave * 4) * 1024) + lane;
            const uint4 rh = yf[0], rl = yf[64], ih = yf[128], il = yf[192];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int gi = h * ROWS + m * 32 + (lane & 31);
                const uint4 ah = lds_hi[gi], al = lds_lo[gi];
                accr[m] = ALM_MFMA(ah, rh, accr[m]);
                acci[m] = ALM_MFMA(ah, ih, acci[m]);
                accr[m] = ALM_MFMA(ah, rl, accr[m]);
                acci[m] = ALM_MFMA(ah, il, acci[m]);
                accr[m] = ALM_MFMA(al, rh, accr[m]);
                acci[m] = ALM_MFMA(al, ih, acci[m]);
            }
        }
        slot = slot + 1 == NSLOT ? 0 : slot + 1;
        nslot = nslot + 1 == NSLOT ? 0 : nslot + 1;
    }
    RIME_MFMA_SETTLE();
    if (c < Ncoeff) {
        float* dst = part + (size_t)split * R * Ncoeff * 2;
        const float iy = 1.0f / y_scale;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = r0 + m * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (row < R) {
                    const float sc = inv_scale[row] * iy;
                    *reinterpret_cast<float2*>(dst + ((size_t)row * Ncoeff + c) * 2) =
                        make_float2(accr[m][e] * sc, -acci[m][e] * sc);
                }
            }
    }
}

// Backward on the packed copy, 8-wave form (MT = 4): one block = 256 coefficients (two packed coefficient tiles) x 128 rows.
// The gout granules a chunk needs are the same for every coefficient tile, so doubling the coefficients per block halves
// their share of the LDS-DMA intake (16 px: Ylm 32 KB + gout 8 KB instead of 16 + 8) -- and that intake, ~21-23 GB/s per
// CU whatever it carries, is what bounds the ring kernels.  Chunks of ONE K step in a ring of four 40-KB
// slots (exactly the 160 KB of a CU: 40 wave instructions per chunk = 5 per wave, no padding loads, no scratch).
struct BwdDma8 {
    static constexpr int ROWS = 128, NW = 8, NSLOT = 4;
    static constexpr int YB = NW * 4 * 1024;           // 8 waves x (re_hi, re_lo, im_hi, im_lo) x 1 KB
    static constexpr int GB = 2 * ROWS * 16;           // one gout image (hi or lo) of the K step: 4 KB
    static constexpr int SLOT = YB + 2 * GB;           // 40 KB
    static constexpr int NGW = GB / 1024;              // 4
    static constexpr int TI = 4 * NW + 2 * NGW;        // 40
    static constexpr int NPW = TI / NW;                // 5
    static constexpr int LDS = NSLOT * SLOT;
    static_assert(TI % NW == 0 && LDS <= 160 * 1024 && NPW * (NSLOT - 2) < 64, "ring shape");
};

__global__ void __launch_bounds__(512, 1)
alm2pix_bwd_packed8_kernel(const uint4* __restrict__ g_hi, const uint4* __restrict__ g_lo,
                           const float* __restrict__ inv_scale, const uint4* __restrict__ Yb,
                           const float* __restrict__ zero16, float y_scale,
                           int R, int Rpad, int Ncoeff, int Npix, int nsteps, int S, int CT2, int RT, float* __restrict__ part)
{
    using C = BwdDma8;
    constexpr int MT = 4, ROWS = C::ROWS, AHEAD = C::NSLOT - 1;
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned smem_addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const int NB = S * CT2 * RT, per_xcd = (NB + 7) / 8;
    const int q = (int)(blockIdx.x % 8) * per_xcd + (int)(blockIdx.x / 8);
    if ((int)(blockIdx.x / 8) >= per_xcd || q >= NB) return;
    const int split = q / (CT2 * RT);
    const int r0 = (q % RT) * ROWS;
    const int ct0 = 2 * ((q / RT) % CT2);              // first of this block's two packed coefficient tiles
    const int c = ct0 * 128 + wave * 32 + (lane & 31);
    const int h = lane >> 5;
    const int send = (Npix + 15) / 16;
    const int nchunk = nsteps;                         // one K step per chunk
    const int niter = split < nchunk ? (nchunk - split + S - 1) / S : 0;
    f32x16 accr[MT], acci[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) { accr[m][e] = 0.f; acci[m][e] = 0.f; }

    const uint4* ytile = Yb + (size_t)ct0 * nsteps * 16 * 64 + lane;
    const size_t ct_stride = (size_t)nsteps * 16 * 64;                 // uint4 elements between packed coefficient tiles
    auto issue = [&](int it, int slot) {
        const int s0 = split + it * S;
        const unsigned base = smem_addr + (unsigned)(slot * C::SLOT);
#pragma unroll
        for (int u = 0; u < C::NPW; ++u) {
            const int t = wave + C::NW * u;            // wave-uniform, 0 .. 39
            if (t < 4 * C::NW) {                       // KB number t of the chunk's Ylm: (tile half, wave, plane)
                glds16(ytile + (size_t)(t >> 4) * ct_stride + ((size_t)s0 * 16 + (t & 15)) * 64, base + t * 1024);
            } else {
                const int tg = t - 4 * C::NW;
                const int img = tg / C::NGW, k = tg % C::NGW;
                const int i = k * 64 + lane;
                const int row = i % ROWS, hh = i / ROWS;
                const int qq = s0 * 2 + hh;
                const uint4* g = img ? g_lo : g_hi;
                const void* src = s0 < send ? (const void*)&g[(size_t)qq * Rpad + r0 + row] : (const void*)zero16;
                glds16(src, base + C::YB + img * C::GB + k * 1024);
            }
        }
    };
#pragma unroll
    for (int a = 0; a < AHEAD; ++a)
        if (a < niter) issue(a, a);
    int slot = 0, nslot = AHEAD;
    for (int it = 0; it < niter; ++it) {
        wait_chunks<C::NPW, AHEAD - 1>(niter - 1 - it);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (it + AHEAD < niter) issue(it + AHEAD, nslot);
        const unsigned char* sl = smem + slot * C::SLOT;
        const uint4* lds_hi = reinterpret_cast<const uint4*>(sl + C::YB);
        const uint4* lds_lo = reinterpret_cast<const uint4*>(sl + C::YB + C::GB);
        const uint4* yf = reinterpret_cast<const uint4*>(sl + (wave * 4) * 1024) + lane;
        const uint4 rh = yf[0], rl = yf[64], ih = yf[128], il = yf[192];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int gi = h * ROWS + m * 32 + (lane & 31);
            const uint4 ah = lds_hi[gi], al = lds_lo[gi];
            accr[m] = ALM_MFMA(ah, rh, accr[m]);
            acci[m] = ALM_MFMA(ah, ih, acci[m]);
            accr[m] = ALM_MFMA(ah, rl, accr[m]);
            acci[m] = ALM_MFMA(ah, il, acci[m]);
            accr[m] = ALM_MFMA(al, rh, accr[m]);
            acci[m] = ALM_MFMA(al, ih, acci[m]);
        }
        slot = slot + 1 == C::NSLOT ? 0 : slot + 1;
        nslot = nslot + 1 == C::NSLOT ? 0 : nslot + 1;
    }
    RIME_MFMA_SETTLE();
    if (c < Ncoeff) {
        float* dst = part + (size_t)split * R * Ncoeff * 2;
        const float iy = 1.0f / y_scale;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = r0 + m * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (row < R) {
                    const float sc = inv_scale[row] * iy;
                    *reinterpret_cast<float2*>(dst + ((size_t)row * Ncoeff + c) * 2) =
                        make_float2(accr[m][e] * sc, -acci[m][e] * sc);
                }
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

// ---- f16-split fast path: shapes and workspace layout ---------------------------------------
struct SplitPlan {
    int MT, Rpad, nsteps;            // row tiles per block, padded rows, K steps of 16
    size_t img_bytes;                // one f16 image (hi or lo) of the row operand
    int S, steps_per_split;          // backward only: pixel splits
};

template <int MT> __global__ void alm2pix_bwd_f16_kernel(const uint4*, const uint4*, const float*, const float*, float,
                                                         int, int, int, int, int, float*);

// blocks of the f16-split backward kernel the chip holds at once (occupancy query, cached per shape)
static long bwd_resident_blocks(int MT)
{
    static long cache[5] = {0, 0, 0, 0, 0};
    if (cache[MT] == 0) {
        int per_cu = 0, dev = 0;
        hipDeviceProp_t prop;
        hipError_t e = hipGetDevice(&dev);
        if (e == hipSuccess) e = hipGetDeviceProperties(&prop, dev);
        if (e == hipSuccess) {
            if (MT == 4) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, alm2pix_bwd_f16_kernel<4>, 256, 0);
            else if (MT == 2) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, alm2pix_bwd_f16_kernel<2>, 256, 0);
            else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, alm2pix_bwd_f16_kernel<1>, 256, 0);
        }
        cache[MT] = (e == hipSuccess && per_cu > 0) ? (long)per_cu * prop.multiProcessorCount : 512;
        (void)hipGetLastError();
    }
    return cache[MT];
}

// LDS-DMA backward kernel: even pixel counts (16-byte aligned Ylm granules); RIME_ALM_BWD_DMA=0 selects the
// register-staged kernel
static bool bwd_use_dma(int Npix)
{
    static int env = -1;
    if (env < 0) { const char* e = getenv("RIME_ALM_BWD_DMA"); env = (e && e[0] == '0') ? 0 : 1; }
    return env == 1 && Npix % 2 == 0;
}

// the LDS-DMA kernel's ring (3 slots of 32 KB Ylm + the gout granules) leaves room for one block per CU
static long bwd_dma_resident_blocks()
{
    static long cache = 0;
    if (cache == 0) {
        int dev = 0; hipDeviceProp_t prop;
        cache = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
        (void)hipGetLastError();
    }
    return cache;
}

static SplitPlan split_plan(int R, int K, int Ncoeff, bool backward, int dma = -1)      // dma: -1 auto, 1 the ring kernels
{
    const bool use_dma = dma < 0 ? bwd_use_dma(K) : dma != 0;
    const bool wide = dma == 2 && R > 64;                          // 8-wave blocks of 256 coefficients, one K step per chunk
    SplitPlan p{};
    p.MT = R > 64 ? 4 : (R > 32 ? 2 : 1);
    p.Rpad = ((R + p.MT * 32 - 1) / (p.MT * 32)) * p.MT * 32;
    p.nsteps = (K + 15) / 16;
    p.img_bytes = (size_t)p.nsteps * 2 * p.Rpad * 16;
    p.S = 1; p.steps_per_split = p.nsteps;
    if (backward) {
        // ~2048 blocks of 128 coefficients x MT row tiles, as WHOLE rounds of the resident grid: every block
        // streams the same number of chunks, so 4.1 rounds cost 5 (C3: 66 tiles x 32 splits = 2112 blocks over
        // 512 resident ones; 31 splits = 2046 blocks = 4.0 rounds)
        const long blocks = (long)(wide ? (Ncoeff + 255) / 256 : (Ncoeff + 127) / 128) * (p.Rpad / (p.MT * 32));
        const long resident = use_dma ? bwd_dma_resident_blocks() : bwd_resident_blocks(p.MT);
        long S = (2048 + blocks - 1) / blocks;                   // measured flat between 1024 and 4096 blocks
        const long maxS = std::max(1, p.nsteps / 32);          // >= 16 chunks of 2 K steps per block
        S = std::max<long>(1, std::min(S, maxS));
        if (use_dma) {
            // LDS-DMA kernel, one block per CU.  Cost of S splits in units of one chunk (2 K steps) of one block:
            // rounds of the chip x (chunks per block + ~3 for the ring's ramp and the epilogue) + the partial plane
            // each split writes and the reduce kernel reads back (R Ncoeff 16 B at ~4 TB/s against ~2.2 us per
            // chunk).  Measured (C3 shape, 66 tiles): S 11 / 15 / 19 / 23 / 31 -> 1.09 / 1.08 / 1.10 / 1.17 / 1.21 ms,
            // S 8 (2.06 rounds) 1.31 ms; 17 tiles x 196608 px: S 15 / 30 / 45 / 60 -> 1.12 / 1.16 / 1.17 / 1.25 ms.
            const long nchunk = wide ? p.nsteps : (p.nsteps + 1) / 2;
            const double c1 = (double)R * Ncoeff * 1.8e-6;
            double best = 1e300;
            const long top = std::min<long>(maxS, 8 * resident / blocks + 1);
            for (long s = 1; s <= std::max<long>(1, top); ++s) {
                const long rounds = (blocks * s + resident - 1) / resident;
                const double cost = (double)rounds * ((nchunk + s - 1) / s + 3) + c1 * s;
                if (cost < best) { best = cost; S = s; }
            }
        } else if (blocks * S > resident) {
            const long rounds = std::max<long>(1, (blocks * S + resident / 2) / resident);    // nearest whole number of rounds
            const long S2 = (rounds * resident) / blocks;                                      // largest split count that fits them
            if (S2 >= 1 && S2 <= maxS) S = S2;
        }
        if (const char* e = getenv("RIME_ALM_BWD_SPLITS")) { const long v = atol(e); if (v >= 1 && v <= maxS) S = v; }   // lab
        p.S = (int)S;
        p.steps_per_split = 0;                                   // chunks are dealt cyclically
    }
    return p;
}

static void launch_split_rows(const float* X, int R, int L, const SplitPlan& p, int neg_odd, void* workspace,
                              hipStream_t st, uint4*& hi, uint4*& lo, float*& inv)
{
    hi = (uint4*)workspace;
    lo = (uint4*)((char*)workspace + p.img_bytes);
    inv = (float*)((char*)workspace + 2 * p.img_bytes);
    unsigned int* rowmax = (unsigned int*)(inv + p.Rpad);
    const int nseg = (2 * p.nsteps * 8 + SPLIT_SEG - 1) / SPLIT_SEG;
    if (p.Rpad % 64 == 0 && (long)R * L >= (1L << 20)) {
        // long rows: scale per row (one block each), then the tiled transposing split; `rowmax` doubles as the scale array
        float* scale = reinterpret_cast<float*>(rowmax);
        (void)hipMemsetAsync((char*)rowmax + (size_t)p.Rpad * sizeof(unsigned int), 0, 64, st);     // the zero granule
        hipLaunchKernelGGL(row_scale_alm_kernel, dim3(p.Rpad), dim3(256), 0, st, X, R, L, scale, inv);
        hipLaunchKernelGGL(split_rows_tiled_kernel, dim3((2 * p.nsteps + 31) / 32, p.Rpad / 64), dim3(256), 0, st, X, R, L,
                           p.Rpad, neg_odd, scale, hi, lo);
        return;
    }
    (void)hipMemsetAsync(rowmax, 0, (size_t)p.Rpad * sizeof(unsigned int) + 64, st);     // + the zero granule behind it
    hipLaunchKernelGGL(row_absmax_kernel, dim3(R, (L + SPLIT_SEG - 1) / SPLIT_SEG), dim3(256), 0, st, X, R, L, rowmax);
    hipLaunchKernelGGL(split_rows_kernel, dim3(p.Rpad, std::max(1, nseg)), dim3(256), 0, st, X, R, L, p.Rpad,
                       neg_odd, rowmax, hi, lo, inv);
}

template <int MT, int KS, int NSLOT>
static hipError_t launch_bwd_dma(dim3 grid, hipStream_t st, const uint4* hi, const uint4* lo, const float* inv, const float* Y,
                                 const float* zero16, float ys, int R, int Rpad, int Ncoeff, int Npix, int S, float* part)
{
    const int CT = (int)grid.y, RT = (int)grid.z;
    const int NB = S * CT * RT;
    grid = dim3(8 * ((NB + 7) / 8));
    constexpr int LDS = BwdDma<MT, KS, NSLOT>::LDS;
    // the attribute belongs to the (function, device) pair: remembered per device of the calling thread
    static unsigned long long configured = 0ull;
    int devid = 0;
    if (hipGetDevice(&devid) != hipSuccess) devid = 0;
    const unsigned long long bit = 1ull << (devid & 63);
    if (!(configured & bit)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&alm2pix_bwd_f16_dma_kernel<MT, KS, NSLOT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return e;
        configured |= bit;
    }
    hipLaunchKernelGGL((alm2pix_bwd_f16_dma_kernel<MT, KS, NSLOT>), grid, dim3(256), LDS, st, hi, lo, inv, Y, zero16, ys, R, Rpad,
                       Ncoeff, Npix, S, CT, RT, part);
    return hipSuccess;
}

template <int MT, int NSLOT>
static hipError_t launch_bwd_packed(dim3 grid, hipStream_t st, const uint4* hi, const uint4* lo, const float* inv, const uint4* Yb,
                                    const float* zero16, float ys, int R, int Rpad, int Ncoeff, int Npix, int nsteps, int S, float* part)
{
    const int CT = (int)grid.y, RT = (int)grid.z;
    const int NB = S * CT * RT;
    grid = dim3(8 * ((NB + 7) / 8));
    constexpr int LDS = BwdDma<MT, PK_BWD_KS, NSLOT>::LDS;
    static unsigned long long configured = 0ull;
    int devid = 0;
    if (hipGetDevice(&devid) != hipSuccess) devid = 0;
    const unsigned long long bit = 1ull << (devid & 63);
    if (!(configured & bit)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&alm2pix_bwd_packed_kernel<MT, NSLOT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return e;
        configured |= bit;
    }
    hipLaunchKernelGGL((alm2pix_bwd_packed_kernel<MT, NSLOT>), grid, dim3(256), LDS, st, hi, lo, inv, Yb, zero16, ys, R, Rpad,
                       Ncoeff, Npix, nsteps, S, CT, RT, part);
    return hipSuccess;
}

static size_t split_ws_bytes(const SplitPlan& p)
{
    return 2 * p.img_bytes + (size_t)p.Rpad * (sizeof(float) + sizeof(unsigned int)) + 64;    // 64: a zero granule
}

} // namespace rime

using namespace rime;

extern "C" size_t rime_alm2pix_fwd_workspace(int dtype, int R, int Ncoeff, int Npix)
{
    if (dtype != RIME_F32 || R <= 0 || Ncoeff <= 0) return 0;
    const SplitPlan p = split_plan(R, 2 * Ncoeff, Ncoeff, false);
    if (Npix <= 0) return split_ws_bytes(p);
    const int S = fwd_packed_splits(R, Ncoeff, Npix, p.MT);                   // rime_alm2pix_fwd_packed: K-split partial planes
    return split_ws_bytes(p) + (S > 1 ? (size_t)S * R * Npix * sizeof(float) : 0);
}

extern "C" int rime_alm2pix_fwd(int dtype, const void* alm, const void* Ylm, double y_scale, int R, int Ncoeff,
                                int Npix, void* out, void* workspace, size_t workspace_bytes, void* stream)
{
    if (!alm || !Ylm || !out || R <= 0 || Ncoeff <= 0 || Npix <= 0 || y_scale < 0) return RIME_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == RIME_F32 && y_scale > 0) {
        const SplitPlan p = split_plan(R, 2 * Ncoeff, Ncoeff, false);
        if (!workspace || workspace_bytes < split_ws_bytes(p)) return RIME_EWORKSPACE;
        uint4 *hi, *lo; float* inv;
        launch_split_rows((const float*)alm, R, 2 * Ncoeff, p, 1, workspace, st, hi, lo, inv);
        dim3 grid((Npix + 127) / 128, p.Rpad / (p.MT * 32));
        const float ys = (float)y_scale;
        const float* Y = (const float*)Ylm; float* o = (float*)out;
        if (p.MT == 4) hipLaunchKernelGGL((alm2pix_fwd_f16_kernel<4>), grid, dim3(256), 0, st, hi, lo, inv, Y, ys, R, p.Rpad, Ncoeff, Npix, o);
        else if (p.MT == 2) hipLaunchKernelGGL((alm2pix_fwd_f16_kernel<2>), grid, dim3(256), 0, st, hi, lo, inv, Y, ys, R, p.Rpad, Ncoeff, Npix, o);
        else hipLaunchKernelGGL((alm2pix_fwd_f16_kernel<1>), grid, dim3(256), 0, st, hi, lo, inv, Y, ys, R, p.Rpad, Ncoeff, Npix, o);
    } else if (dtype == RIME_F32) {
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
    if (dtype != RIME_F32 || R <= 0 || Ncoeff <= 0 || Npix <= 0) return 0;
    // exact-f32 path: split partials; f16-split path: images + row scales + split partials
    const int S0 = bwd_splits(R, Ncoeff, Npix);
    const size_t exact = S0 <= 1 ? 0 : (size_t)S0 * R * Ncoeff * 2 * sizeof(float);
    const SplitPlan p = split_plan(R, Npix, Ncoeff, true);
    const size_t fast = split_ws_bytes(p) + (size_t)p.S * R * Ncoeff * 2 * sizeof(float);
    const SplitPlan pp = split_plan(R, Npix, Ncoeff, true, 1);                 // rime_alm2pix_bwd_packed: always a ring
    const SplitPlan pw = split_plan(R, Npix, Ncoeff, true, 2);
    const size_t packed = split_ws_bytes(pp) + (size_t)std::max(pp.S, pw.S) * R * Ncoeff * 2 * sizeof(float);
    return std::max(exact, std::max(fast, packed));
}

extern "C" int rime_alm2pix_bwd(int dtype, const void* gout, const void* Ylm, double y_scale, int R, int Ncoeff,
                                int Npix, void* galm, void* workspace, size_t workspace_bytes,
                                void* stream)
{
    if (!gout || !Ylm || !galm || R <= 0 || Ncoeff <= 0 || Npix <= 0 || y_scale < 0) return RIME_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == RIME_F32 && y_scale > 0) {
        const SplitPlan p = split_plan(R, Npix, Ncoeff, true);
        const size_t len = (size_t)R * Ncoeff * 2;
        const size_t need = split_ws_bytes(p) + (size_t)p.S * len * sizeof(float);
        if (!workspace || workspace_bytes < need) return RIME_EWORKSPACE;
        uint4 *hi, *lo; float* inv;
        launch_split_rows((const float*)gout, R, Npix, p, 0, workspace, st, hi, lo, inv);
        float* part = p.S > 1 ? (float*)((char*)workspace + split_ws_bytes(p)) : (float*)galm;
        dim3 grid(p.S, (Ncoeff + 127) / 128, p.Rpad / (p.MT * 32));
        const float ys = (float)y_scale;
        const float* Y = (const float*)Ylm;
        if (bwd_use_dma(Npix)) {
            const float* zero16 = (const float*)((char*)workspace + split_ws_bytes(p) - 64);
            hipError_t e = hipSuccess;
            // chunks of 2 K steps in a ring of 3 slots; 1 K step in 6 slots (more, smaller loads in flight) measured the same
            if (p.MT == 4) e = launch_bwd_dma<4, 2, 3>(grid, st, hi, lo, inv, Y, zero16, ys, R, p.Rpad, Ncoeff, Npix, p.S, part);
            else if (p.MT == 2) e = launch_bwd_dma<2, 2, 3>(grid, st, hi, lo, inv, Y, zero16, ys, R, p.Rpad, Ncoeff, Npix, p.S, part);
            else e = launch_bwd_dma<1, 2, 3>(grid, st, hi, lo, inv, Y, zero16, ys, R, p.Rpad, Ncoeff, Npix, p.S, part);
            if (e != hipSuccess) return RIME_ELAUNCH;
        }
        else if (p.MT == 4) hipLaunchKernelGGL((alm2pix_bwd_f16_kernel<4>), grid, dim3(256), 0, st, hi, lo, inv, Y, ys, R, p.Rpad, Ncoeff, Npix, p.S, part);
        else if (p.MT == 2) hipLaunchKernelGGL((alm2pix_bwd_f16_kernel<2>), grid, dim3(256), 0, st, hi, lo, inv, Y, ys, R, p.Rpad, Ncoeff, Npix, p.S, part);
        else hipLaunchKernelGGL((alm2pix_bwd_f16_kernel<1>), grid, dim3(256), 0, st, hi, lo, inv, Y, ys, R, p.Rpad, Ncoeff, Npix, p.S, part);
        if (p.S > 1) {
            int nb = (int)std::min<size_t>((len + 255) / 256, 2048);
            hipLaunchKernelGGL(alm_reduce_kernel, dim3(nb), dim3(256), 0, st, part, (float*)galm, len, p.S);
        }
        return check_launch();
    }
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
    if (dtype != RIME_F64) return RIME_EINVAL;
    constexpr int CT = 4, RTB = 8;
    dim3 grid((Ncoeff + CT - 1) / CT, (R + RTB - 1) / RTB);
    hipLaunchKernelGGL((alm2pix_bwd_kernel<double, CT, RTB>), grid, dim3(256), 0, st,
                       (const double*)gout, (const double*)Ylm, R, Ncoeff, Npix, (double*)galm);
    return check_launch();
}


// ---- packed Ylm (see the kernels): size, packing, and the two transforms on a packed copy ------------------------
extern "C" size_t rime_alm2pix_packed_bytes(int Ncoeff, int Npix, int direction)
{
    if (Ncoeff <= 0 || Npix <= 0 || (direction != 0 && direction != 1)) return 0;
    if (direction == 0) return (size_t)((Npix + 127) / 128 * 4) * pk_fwd_steps(Ncoeff) * 2048;     // whole 4-wave blocks of pixel tiles
    return (size_t)((Ncoeff + 255) / 256 * 2) * pk_bwd_steps(Npix) * 16384;      // whole 256-coefficient (8-wave) blocks
}

extern "C" int rime_alm2pix_pack(const void* Ylm, double y_scale, int Ncoeff, int Npix, int direction, void* packed,
                                 void* stream)
{
    if (!Ylm || !packed || Ncoeff <= 0 || Npix <= 0 || !(y_scale > 0) || (direction != 0 && direction != 1)) return RIME_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (direction == 0) {
        const int nsteps = pk_fwd_steps(Ncoeff);
        const size_t ntile = (size_t)(Npix + 127) / 128 * 4;
        const size_t nthr = ntile * nsteps * 64;
        if ((nthr + 255) / 256 > 0x7fffffffull) return RIME_EUNSUPPORTED;
        hipLaunchKernelGGL(alm_pack_fwd_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, st, (const float*)Ylm, (float)y_scale,
                           Ncoeff, Npix, nsteps, ntile, (uint4*)packed);
    } else {
        const int nsteps = pk_bwd_steps(Npix);
        const size_t nblk = (size_t)((Ncoeff + 255) / 256 * 2) * nsteps;
        if (nblk > 0x7fffffffull) return RIME_EUNSUPPORTED;
        hipLaunchKernelGGL(alm_pack_bwd_kernel, dim3((unsigned)nblk), dim3(256), 0, st, (const float*)Ylm, (float)y_scale,
                           Ncoeff, Npix, nsteps, (uint4*)packed);
    }
    return check_launch();
}

extern "C" int rime_alm2pix_fwd_packed(const void* alm, const void* packed, double y_scale, int R, int Ncoeff, int Npix,
                                       void* out, void* workspace, size_t workspace_bytes, void* stream)
{
    if (!alm || !packed || !out || R <= 0 || Ncoeff <= 0 || Npix <= 0 || !(y_scale > 0)) return RIME_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const SplitPlan p = split_plan(R, 2 * Ncoeff, Ncoeff, false);
    if (!workspace || workspace_bytes < split_ws_bytes(p)) return RIME_EWORKSPACE;
    uint4 *hi, *lo; float* inv;
    launch_split_rows((const float*)alm, R, 2 * Ncoeff, p, 1, workspace, st, hi, lo, inv);
    const int S = fwd_packed_splits(R, Ncoeff, Npix, p.MT);
    const size_t len = (size_t)R * Npix;
    if (S > 1 && workspace_bytes < split_ws_bytes(p) + (size_t)S * len * sizeof(float)) return RIME_EWORKSPACE;
    dim3 grid((Npix + 127) / 128, p.Rpad / (p.MT * 32), S);
    const float ys = (float)y_scale;
    const uint4* Y = (const uint4*)packed;
    float* o = S > 1 ? (float*)((char*)workspace + split_ws_bytes(p)) : (float*)out;
    const int nsteps = pk_fwd_steps(Ncoeff);
    const int cps = (nsteps / PK_FWD_CHUNK + S - 1) / S;
    if (p.MT == 4) hipLaunchKernelGGL((alm2pix_fwd_packed_kernel<4>), grid, dim3(256), 0, st, hi, lo, inv, Y, ys, R, p.Rpad, Ncoeff, Npix, nsteps, cps, o);
    else if (p.MT == 2) hipLaunchKernelGGL((alm2pix_fwd_packed_kernel<2>), grid, dim3(256), 0, st, hi, lo, inv, Y, ys, R, p.Rpad, Ncoeff, Npix, nsteps, cps, o);
    else hipLaunchKernelGGL((alm2pix_fwd_packed_kernel<1>), grid, dim3(256), 0, st, hi, lo, inv, Y, ys, R, p.Rpad, Ncoeff, Npix, nsteps, cps, o);
    if (S > 1) {
        int nb = (int)std::min<size_t>((len + 255) / 256, 2048);
        hipLaunchKernelGGL(alm_reduce_kernel, dim3(nb), dim3(256), 0, st, o, (float*)out, len, S);
    }
    return check_launch();
}

extern "C" int rime_alm2pix_bwd_packed(const void* gout, const void* packed, double y_scale, int R, int Ncoeff, int Npix,
                                       void* galm, void* workspace, size_t workspace_bytes, void* stream)
{
    if (!gout || !packed || !galm || R <= 0 || Ncoeff <= 0 || Npix <= 0 || !(y_scale > 0)) return RIME_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    static int wide_env = -1;
    if (wide_env < 0) { const char* ev = getenv("RIME_ALM_BWD_WIDE"); wide_env = (ev && ev[0] == '0') ? 0 : 1; }
    const bool wide = wide_env == 1 && R > 64;
    const SplitPlan p = split_plan(R, Npix, Ncoeff, true, wide ? 2 : 1);
    const size_t len = (size_t)R * Ncoeff * 2;
    const size_t need = split_ws_bytes(p) + (size_t)p.S * len * sizeof(float);
    if (!workspace || workspace_bytes < need) return RIME_EWORKSPACE;
    uint4 *hi, *lo; float* inv;
    launch_split_rows((const float*)gout, R, Npix, p, 0, workspace, st, hi, lo, inv);
    float* part = p.S > 1 ? (float*)((char*)workspace + split_ws_bytes(p)) : (float*)galm;
    dim3 grid(p.S, (Ncoeff + 127) / 128, p.Rpad / (p.MT * 32));
    const float* zero16 = (const float*)((char*)workspace + split_ws_bytes(p) - 64);
    const uint4* Y = (const uint4*)packed;
    const int nsteps = pk_bwd_steps(Npix);
    hipError_t e;
    if (wide) {
        const int CT2 = (Ncoeff + 255) / 256, RT = p.Rpad / 128, NB = p.S * CT2 * RT;
        static unsigned long long configured = 0ull;
        int devid = 0;
        if (hipGetDevice(&devid) != hipSuccess) devid = 0;
        const unsigned long long bit = 1ull << (devid & 63);
        e = hipSuccess;
        if (!(configured & bit)) {
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&alm2pix_bwd_packed8_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, BwdDma8::LDS);
            if (e == hipSuccess) configured |= bit;
        }
        if (e == hipSuccess)
            hipLaunchKernelGGL(alm2pix_bwd_packed8_kernel, dim3(8 * ((NB + 7) / 8)), dim3(512), BwdDma8::LDS, st, hi, lo, inv, Y,
                               zero16, (float)y_scale, R, p.Rpad, Ncoeff, Npix, nsteps, p.S, CT2, RT, part);
    }
    else if (p.MT == 4) e = launch_bwd_packed<4, 3>(grid, st, hi, lo, inv, Y, zero16, (float)y_scale, R, p.Rpad, Ncoeff, Npix, nsteps, p.S, part);
    else if (p.MT == 2) e = launch_bwd_packed<2, 3>(grid, st, hi, lo, inv, Y, zero16, (float)y_scale, R, p.Rpad, Ncoeff, Npix, nsteps, p.S, part);
    else e = launch_bwd_packed<1, 3>(grid, st, hi, lo, inv, Y, zero16, (float)y_scale, R, p.Rpad, Ncoeff, Npix, nsteps, p.S, part);
    if (e != hipSuccess) return RIME_ELAUNCH;
    if (p.S > 1) {
        int nb = (int)std::min<size_t>((len + 255) / 256, 2048);
        hipLaunchKernelGGL(alm_reduce_kernel, dim3(nb), dim3(256), 0, st, part, (float*)galm, len, p.S);
    }
    return check_launch();
}
