// fringe_mfma.hip -- antenna-factored fringe sum on the matrix cores (gfx950), forward.
//
// For baselines that are antenna pairs, b_ij = r_j - r_i, the fringe factorises:
//     exp(2 pi i nu b_ij.s / c) = E_j conj(E_i),   E_a[f,p] = exp(2 pi i nu_f r_a.s_p / c)
// so for every (time, channel) the visibilities of ALL pairs are one Hermitian rank-P update
//     V[i,j] = sum_p conj(E_i[p]) * (A[p] E_j[p]),        A = psky[t,f,:]  (1-pol, real)
// i.e. a complex GEMM with M = N = Nant, K = P, batched over (t, f) -- the "dense
// (Nvis x Npix) . Npix contraction" of the north star, at 1/Nant of the exponentials of the
// baseline formulation.  (Reference arithmetic replaced: the same lines as fringe.hip,
// telescope_model.py:310-358 + rime_model.py:423-429.)
//
// Precision: f32 inputs are split into two f16 halves (hi + lo, 21 significant bits); the three
// cross products hi*hi + hi*lo + lo*hi run on v_mfma_f32_32x32x16_f16 with f32 accumulation
// (the dropped lo*lo term is 2^-22 relative).  psky rows are pre-scaled by a power of two per
// (t, f) so that the f16 range is used (`scale` input); accumulators are flushed to memory every
// 2048 pixels so f32 accumulation error stays ~eps*sqrt(128) per flush.
//
// Work decomposition: block = one (t, f [, pixel split]); 4 waves; the upper-triangular 32x32
// tiles of the Nant x Nant (<= 128 x 128) output are dealt to the waves.  Per panel of 16
// pixels the block generates E for all antennas once (f64 delay + phase reduction, hardware
// sin/cos), writes the f16 hi/lo operand images to LDS ([antenna][pixel][re,im], 16-B fragment
// granules, rows padded to 80 B: conflict-free ds_read_b128), and every wave runs its MFMAs from
// LDS fragments while the next panel is generated into the other buffer.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include "rime_common.h"

namespace rime {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr int MF_NA = 128;                   // antennas per block (4 x 4 tiles)
constexpr int MF_SPLIT_PIX = 8192;           // pixels per block (bounds the f32 MFMA accumulation chain)

struct AntArgs {
    const double* antpos;      // [Nant, 3]
    const double* sdir;        // [Nt, 3, Pstride]
    const double* freqs;       // [Nf]
    const float* psky;         // strided [t][f][p]
    const float* scale;        // [Nt, Nf] power-of-two pre-scale of psky rows
    const int* pair_direct;    // [128*128] baseline slot receiving V[i,j], or -1
    const int* pair_conj;      // [128*128] baseline slot receiving conj(V[i,j]), or -1
    float* vis;                // [Nbl, Nt, Nf, 2]
    float* ws;                 // partial slabs when S > 1
    int Nant, Nbl, Nt, Nf, Pstride;
    int S, panels_per_split;
    long long st_t, st_f;
    double sign;
};

__device__ __forceinline__ uint32_t pack_rtz(float a, float b)
{
    auto h = __builtin_amdgcn_cvt_pkrtz(a, b);
    return __builtin_bit_cast(uint32_t, h);
}

// split (a, b) into f16 hi and lo pairs: x = hi + lo + O(2^-21 |x|)
__device__ __forceinline__ void split2(float a, float b, uint32_t& hi, uint32_t& lo)
{
    auto h = __builtin_amdgcn_cvt_pkrtz(a, b);
    hi = __builtin_bit_cast(uint32_t, h);
    const float ra = a - (float)h[0];
    const float rb = b - (float)h[1];
    lo = pack_rtz(ra, rb);
}

__device__ __forceinline__ f16x8 as_frag(const uint4& v) { return __builtin_bit_cast(f16x8, v); }

// (Lr, Li) -> (-Li, Lr) for each of the four packed pairs: left operand of the imaginary part
__device__ __forceinline__ uint4 rot90(const uint4& v)
{
    uint4 r;
    r.x = __builtin_amdgcn_alignbit(v.x, v.x, 16) ^ 0x00008000u;
    r.y = __builtin_amdgcn_alignbit(v.y, v.y, 16) ^ 0x00008000u;
    r.z = __builtin_amdgcn_alignbit(v.z, v.z, 16) ^ 0x00008000u;
    r.w = __builtin_amdgcn_alignbit(v.w, v.w, 16) ^ 0x00008000u;
    return r;
}

// KP = pixels per panel (16 = two MFMA K-steps of 8 pixels).  LDS rows are KP*4 + 16 bytes, an odd
// number of 16-B granules, so the 16 lanes of a ds_read_b128 group hit 16 distinct granules.
// One LDS buffer (4 images, 40 KB) and two barriers per panel: several blocks are resident per CU,
// so one block's operand generation (VALU) runs under another block's MFMAs.
// Every block covers at most MF_SPLIT_PIX pixels and STORES its result (no read-modify-write):
// f32 accumulation inside the MFMA chain stays below eps*sqrt(512/2), and the pixel splits are
// summed by reduce_vis_kernel in a fixed order (deterministic).
// Measured at C4 (profiles/r01): 13.7 ms = 8.9 ms with the generator disabled (matrix pipe ~94 %
// busy) + 6.1 ms of generation -- the two phases do not overlap: co-resident blocks run the loop in
// lockstep.  Variants tried and measured NOT faster (kept out of the tree for simplicity):
// 4 MFMA-only + 4 generator waves per block with double-buffered images (15.8 ms); symmetric
// sqrt|psky| weighting with one image pair + sign masks, double-buffered (14.1 ms); generator
// pieces interleaved with the MFMA groups in program order (13.8 ms); f32 instead of f64 phase
// arithmetic (13.5 ms, numerically wrong: shows the f64 ops are not the limiter).
template <int MF_KP>
__global__ void __launch_bounds__(256, 2)
fringe_ant_fwd_kernel(AntArgs A)
{
    constexpr int MF_ROWB = MF_KP * 4 + 16;
    constexpr int MF_IMG = MF_NA * MF_ROWB;
    constexpr int GPL = MF_KP;                  // lanes per pixel group in the generation mapping
    constexpr int APT = MF_NA * MF_KP / 256;    // antennas per thread per panel
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char* img = smem;                                         // 4 images: L hi, L lo, B hi, B lo
    double* ant_lds = reinterpret_cast<double*>(smem + 4 * MF_IMG);    // [128][3]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int f = blockIdx.y;
    const int t = blockIdx.z / A.S, split = blockIdx.z % A.S;
    const int TA = (A.Nant + 31) / 32;

    for (int i = tid; i < MF_NA * 3; i += 256)
        ant_lds[i] = (i < A.Nant * 3) ? A.sign * A.antpos[i] : 0.0;

    const double nu_c = A.freqs[f] * (1.0 / 2.99792458e8);
    const float scl = A.scale[t * A.Nf + f];
    const float* arow = A.psky + (size_t)t * A.st_t + (size_t)f * A.st_f;
    const double* sd = A.sdir + (size_t)t * 3 * A.Pstride;

    // this wave's tiles: w, w+4, w+8 of the row-major upper-triangle enumeration
    int ti[3], tj[3], nt = 0;
    {
        int idx = 0;
        for (int a = 0; a < TA; ++a)
            for (int b = a; b < TA; ++b, ++idx)
                if ((idx & 3) == wave && nt < 3) { ti[nt] = a; tj[nt] = b; ++nt; }
    }
    f32x16 accR[3], accI[3];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int e = 0; e < 16; ++e) { accR[q][e] = 0.f; accI[q][e] = 0.f; }

    const int npanel = A.Pstride / MF_KP;
    const int pbeg = split * A.panels_per_split;
    const int pend = min(npanel, pbeg + A.panels_per_split);

    // generation mapping: 16 consecutive lanes = the 16 pixels of the panel; a thread handles its
    // pixel for APT consecutive antennas
    const int gp = tid & (GPL - 1), ga0 = (tid / GPL) * APT;
    unsigned char* gbase = img + gp * 4 + ga0 * MF_ROWB;
    const int koff = 4 * (lane >> 5) * 4;                              // this lane half's 4 pixels
    int roff[3], coff[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        roff[q] = (ti[q < nt ? q : 0] * 32 + (lane & 31)) * MF_ROWB + koff;
        coff[q] = (tj[q < nt ? q : 0] * 32 + (lane & 31)) * MF_ROWB + koff + 2 * MF_IMG;
    }

    for (int panel = pbeg; panel < pend; ++panel) {
        __syncthreads();                               // previous panel's fragments consumed
        {
            const int p = panel * MF_KP + gp;
            // pointing vector pre-multiplied by nu/c: the antenna dot product is the phase in turns
            const double ux = sd[p] * nu_c, uy = sd[A.Pstride + p] * nu_c, uz = sd[2 * (size_t)A.Pstride + p] * nu_c;
            const float a = arow[p] * scl;
#pragma unroll 2
            for (int u = 0; u < APT; ++u) {
                const int an = ga0 + u;
                const double ph = ant_lds[3 * an] * ux + ant_lds[3 * an + 1] * uy + ant_lds[3 * an + 2] * uz;
                const float r = (float)(ph - rint(ph));
                const float s = __builtin_amdgcn_sinf(r), c = __builtin_amdgcn_cosf(r);
                uint32_t hi, lo;
                split2(c, s, hi, lo);
                *reinterpret_cast<uint32_t*>(gbase + 0 * MF_IMG + u * MF_ROWB) = hi;
                *reinterpret_cast<uint32_t*>(gbase + 1 * MF_IMG + u * MF_ROWB) = lo;
                split2(a * c, a * s, hi, lo);
                *reinterpret_cast<uint32_t*>(gbase + 2 * MF_IMG + u * MF_ROWB) = hi;
                *reinterpret_cast<uint32_t*>(gbase + 3 * MF_IMG + u * MF_ROWB) = lo;
            }
        }
        __syncthreads();
        {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                if (q < nt) {
#pragma unroll
                    for (int ks = 0; ks < MF_KP / 8; ++ks) {
                        const uint4 Lh = *reinterpret_cast<const uint4*>(img + roff[q] + ks * 32);
                        const uint4 Ll = *reinterpret_cast<const uint4*>(img + roff[q] + ks * 32 + MF_IMG);
                        const uint4 Bh = *reinterpret_cast<const uint4*>(img + coff[q] + ks * 32);
                        const uint4 Bl = *reinterpret_cast<const uint4*>(img + coff[q] + ks * 32 + MF_IMG);
                        const f16x8 lh = as_frag(Lh), ll = as_frag(Ll), bh = as_frag(Bh), bl = as_frag(Bl);
                        const f16x8 lh2 = as_frag(rot90(Lh)), ll2 = as_frag(rot90(Ll));
                        accR[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(lh, bh, accR[q], 0, 0, 0);
                        accI[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(lh2, bh, accI[q], 0, 0, 0);
                        accR[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(lh, bl, accR[q], 0, 0, 0);
                        accI[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(lh2, bl, accI[q], 0, 0, 0);
                        accR[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ll, bh, accR[q], 0, 0, 0);
                        accI[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ll2, bh, accI[q], 0, 0, 0);
                    }
                }
            }
        }
    }

    // epilogue: V[i,j] / scale -> the baseline slot(s) of pair (i,j); one store per element
    const size_t vis_elems = (size_t)A.Nbl * A.Nt * A.Nf * 2;
    float* dst = (A.S == 1) ? A.vis : A.ws + (size_t)split * vis_elems;
    const float inv = 1.0f / scl;
    const int col = lane & 31;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        if (q < nt) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                const int i = ti[q] * 32 + row, j = tj[q] * 32 + col;
                const float vr = accR[q][e] * inv, vi = accI[q][e] * inv;
                const int bd = A.pair_direct[i * MF_NA + j];
                if (bd >= 0) {
                    float2* o = reinterpret_cast<float2*>(dst + (((size_t)bd * A.Nt + t) * A.Nf + f) * 2);
                    *o = make_float2(vr, vi);
                }
                const int bc = A.pair_conj[i * MF_NA + j];
                if (bc >= 0) {
                    float2* o = reinterpret_cast<float2*>(dst + (((size_t)bc * A.Nt + t) * A.Nf + f) * 2);
                    *o = make_float2(vr, -vi);
                }
            }
        }
    }
}


// ---------------------------------------------------------------------------------------
// backward:  gpsky[t,f,p] = Re sum_{i,j} E_i(p) conj(E_j(p)) G[i,j]
//                        = sum_i ( Er_i Tr_i + Ei_i Ti_i ),   T_i(p) = sum_j conj(G[i,j]) E_j(p)
// G[i,j] (tile(i) <= tile(j)) collects gvis of pair (i -> j) and the conjugate of pair (j -> i)
// through the same two tables as the forward.  T is a block-upper-triangular complex GEMM
// (M = antennas i, N = pixels, K = antennas j) on v_mfma_f32_32x32x16_f16 with the same hi/lo
// f16 split; G (scaled by a power of two per (t,f)) sits in LDS in A-fragment order for the whole
// block, E fragments are generated in registers by the lane that consumes them (lane = pixel, so
// every E value is computed exactly once), and the final contraction with E_i is lane-local:
// the D rows a lane holds are exactly the antennas it generated.  No atomics.
// ---------------------------------------------------------------------------------------
struct AntBwdArgs {
    const double* antpos; const double* sdir; const double* freqs;
    const float* gvis;         // [Nbl, Nt, Nf, 2]
    const float* gscale;       // [Nt, Nf] power-of-two pre-scale of gvis
    const int* pair_direct; const int* pair_conj;
    float* gpsky;              // strided [t][f][p]
    int Nant, Nbl, Nt, Nf, Pstride;
    int S, tiles_per_split;    // pixel tiles (32 px) per block
    long long st_t, st_f;
    double sign;
};

constexpr int MB_TILES = 10;                       // upper-triangular 32x32 tiles of a 128x128 matrix
constexpr int MB_GIMG = MB_TILES * 4 * 2 * 32 * 16;  // bytes per G image (hi or lo): 40960

__device__ __forceinline__ int tri_index(int ti, int tj) { return ti * 4 - ti * (ti - 1) / 2 + (tj - ti); }

__global__ void __launch_bounds__(512, 2)
fringe_ant_bwd_kernel(AntBwdArgs A)
{
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char* g_img = smem;                                          // 2 images (hi, lo)
    double* ant_lds = reinterpret_cast<double*>(smem + 2 * MB_GIMG);      // [128][3]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int f = blockIdx.y;
    const int t = blockIdx.z / A.S, split = blockIdx.z % A.S;
    const int TA = (A.Nant + 31) / 32;

    for (int i = tid; i < MF_NA * 3; i += 512)
        ant_lds[i] = (i < A.Nant * 3) ? A.sign * A.antpos[i] : 0.0;

    // stage G: element (tile, ks, h, row, r) <-> pair (i = 32 ti + row, j = 32 tj + 8 ks + 4 h + r)
    const float gs = A.gscale[t * A.Nf + f];
    for (int e = tid; e < MB_TILES * 4 * 2 * 32 * 4; e += 512) {
        const int r = e & 3, row = (e >> 2) & 31, h = (e >> 7) & 1, ks = (e >> 8) & 3, tile = e >> 10;
        int ti = 0, rem = tile;
        while (rem >= 4 - ti) { rem -= 4 - ti; ++ti; }
        const int tj = ti + rem;
        const int i = 32 * ti + row, j = 32 * tj + 8 * ks + 4 * h + r;
        float gr = 0.f, gi = 0.f;
        if (ti < TA && tj < TA) {
            const int bd = A.pair_direct[i * MF_NA + j];
            if (bd >= 0) {
                const float* g = A.gvis + (((size_t)bd * A.Nt + t) * A.Nf + f) * 2;
                gr += g[0]; gi += g[1];
            }
            const int bc = A.pair_conj[i * MF_NA + j];
            if (bc >= 0) {
                const float* g = A.gvis + (((size_t)bc * A.Nt + t) * A.Nf + f) * 2;
                gr += g[0]; gi -= g[1];
            }
        }
        uint32_t hi, lo;
        split2(gr * gs, gi * gs, hi, lo);
        const int off = ((((tile * 4 + ks) * 2 + h) * 32 + row) * 4 + r) * 4;
        *reinterpret_cast<uint32_t*>(g_img + off) = hi;
        *reinterpret_cast<uint32_t*>(g_img + MB_GIMG + off) = lo;
    }
    __syncthreads();

    const double nu_c = A.freqs[f] * (1.0 / 2.99792458e8);
    const double* sd = A.sdir + (size_t)t * 3 * A.Pstride;
    float* orow = A.gpsky + (size_t)t * A.st_t + (size_t)f * A.st_f;
    const float inv = 1.0f / gs;
    const int h = lane >> 5;
    const int ntile = A.Pstride / 32;
    const int tbeg = split * A.tiles_per_split;
    const int tend = min(ntile, tbeg + A.tiles_per_split);

    for (int pt = tbeg + wave; pt < tend; pt += 8) {
        const int p = pt * 32 + (lane & 31);
        const double sx = sd[p] * nu_c, sy = sd[A.Pstride + p] * nu_c, sz = sd[2 * (size_t)A.Pstride + p] * nu_c;
        f32x16 accR[4], accI[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 16; ++e) { accR[q][e] = 0.f; accI[q][e] = 0.f; }
        float part = 0.f;
#pragma unroll
        for (int tjr = 0; tjr < 4; ++tjr) {
            const int tj = 3 - tjr;                          // descending: row tile tj completes here
            if (tj < TA) {
                float ec[4][4], es[4][4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    uint4 Eh, El;
                    uint32_t* eh = reinterpret_cast<uint32_t*>(&Eh);
                    uint32_t* el = reinterpret_cast<uint32_t*>(&El);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int an = 32 * tj + 8 * ks + 4 * h + r;
                        const double ph = ant_lds[3 * an] * sx + ant_lds[3 * an + 1] * sy + ant_lds[3 * an + 2] * sz;
                        const float rr = (float)(ph - rint(ph));
                        const float s = __builtin_amdgcn_sinf(rr), c = __builtin_amdgcn_cosf(rr);
                        ec[ks][r] = c; es[ks][r] = s;
                        split2(c, s, eh[r], el[r]);
                    }
                    const f16x8 bh = as_frag(Eh), bl = as_frag(El);
#pragma unroll
                    for (int ti = 0; ti < 4; ++ti) {
                        if (ti <= tj) {
                            const int off = (((tri_index(ti, tj) * 4 + ks) * 2 + h) * 32 + (lane & 31)) * 16;
                            const uint4 Gh = *reinterpret_cast<const uint4*>(g_img + off);
                            const uint4 Gl = *reinterpret_cast<const uint4*>(g_img + MB_GIMG + off);
                            const f16x8 gh = as_frag(Gh), gl = as_frag(Gl);
                            const f16x8 gh2 = as_frag(rot90(Gh)), gl2 = as_frag(rot90(Gl));
                            accR[ti] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh, bh, accR[ti], 0, 0, 0);
                            accI[ti] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh2, bh, accI[ti], 0, 0, 0);
                            accR[ti] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh, bl, accR[ti], 0, 0, 0);
                            accI[ti] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gh2, bl, accI[ti], 0, 0, 0);
                            accR[ti] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gl, bh, accR[ti], 0, 0, 0);
                            accI[ti] = __builtin_amdgcn_mfma_f32_32x32x16_f16(gl2, bh, accI[ti], 0, 0, 0);
                        }
                    }
                }
                // row tile tj is complete: contract with E_i of the same antennas (lane-local)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    part = fmaf(ec[e >> 2][e & 3], accR[tj][e], part);
                    part = fmaf(es[e >> 2][e & 3], accI[tj][e], part);
                }
            }
        }
        part += __shfl_xor(part, 32, 64);
        if (h == 0) orow[p] = part * inv;
    }
}

template <typename T>
__global__ void reduce_vis_kernel(const T* __restrict__ ws, T* __restrict__ out, size_t len, int S)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += (size_t)gridDim.x * blockDim.x) {
        T v = T(0);
        for (int s = 0; s < S; ++s) v += ws[(size_t)s * len + i];
        out[i] = v;
    }
}

static int ant_splits(int Nt, int Nf, int Pstride)
{
    // at most MF_SPLIT_PIX pixels per block; more splits while the grid is below ~4 blocks per CU
    long S = (Pstride + MF_SPLIT_PIX - 1) / MF_SPLIT_PIX;
    const long blocks = (long)Nt * Nf;
    const long maxS = std::max(1, Pstride / 1024);
    while (blocks * S < 1024 && S < maxS) ++S;
    return (int)std::max<long>(1, S);
}

} // namespace rime

using namespace rime;

extern "C" size_t rime_fringe_ant_workspace(int Nbl, int Nt, int Nf, int Pstride)
{
    const int S = ant_splits(Nt, Nf, Pstride);
    return S <= 1 ? 0 : (size_t)S * Nbl * Nt * Nf * 2 * sizeof(float);
}

extern "C" int rime_fringe_ant_fwd(const double* antpos, const double* sdir, const double* freqs,
                                   const float* psky, const float* scale, const int* pair_direct,
                                   const int* pair_conj, int Nant, int Nbl, int Nt, int Nf, int Pstride,
                                   long long st_t, long long st_f, int sign, float* vis,
                                   void* workspace, size_t workspace_bytes, void* stream)
{
    if (!antpos || !sdir || !freqs || !psky || !scale || !pair_direct || !pair_conj || !vis) return RIME_EINVAL;
    if (Nant <= 0 || Nant > MF_NA || Nbl <= 0 || Nt <= 0 || Nf <= 0 || Pstride <= 0 || Pstride % 64 != 0)
        return RIME_EINVAL;
    if (sign != 1 && sign != -1) return RIME_EINVAL;
    AntArgs A{};
    A.antpos = antpos; A.sdir = sdir; A.freqs = freqs; A.psky = psky; A.scale = scale;
    A.pair_direct = pair_direct; A.pair_conj = pair_conj; A.vis = vis; A.ws = (float*)workspace;
    A.Nant = Nant; A.Nbl = Nbl; A.Nt = Nt; A.Nf = Nf; A.Pstride = Pstride;
    A.st_t = st_t; A.st_f = st_f; A.sign = (double)sign;
    constexpr int KP = 16;
    A.S = ant_splits(Nt, Nf, Pstride);
    const int npanel = Pstride / KP;
    A.panels_per_split = (npanel + A.S - 1) / A.S;
    A.panels_per_split = ((A.panels_per_split + 3) / 4) * 4;        // splits start on 64-pixel tiles
    A.S = (npanel + A.panels_per_split - 1) / A.panels_per_split;
    const size_t vis_elems = (size_t)Nbl * Nt * Nf * 2;
    if (A.S > 1 && workspace_bytes < (size_t)A.S * vis_elems * sizeof(float)) return RIME_EWORKSPACE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const size_t lds = 4 * (size_t)MF_NA * (KP * 4 + 16) + MF_NA * 3 * sizeof(double);
    dim3 grid(1, Nf, Nt * A.S);
    hipLaunchKernelGGL((fringe_ant_fwd_kernel<KP>), grid, dim3(256), lds, st, A);
    if (A.S > 1) {
        int nb = (int)std::min<size_t>((vis_elems + 255) / 256, 4096);
        hipLaunchKernelGGL((reduce_vis_kernel<float>), dim3(nb), dim3(256), 0, st, A.ws, vis, vis_elems, A.S);
    }
    return check_launch();
}

extern "C" int rime_fringe_ant_bwd(const double* antpos, const double* sdir, const double* freqs,
                                   const float* gvis, const float* gscale, const int* pair_direct,
                                   const int* pair_conj, int Nant, int Nbl, int Nt, int Nf, int Pstride,
                                   long long st_t, long long st_f, int sign, float* gpsky, void* stream)
{
    if (!antpos || !sdir || !freqs || !gvis || !gscale || !pair_direct || !pair_conj || !gpsky) return RIME_EINVAL;
    if (Nant <= 0 || Nant > MF_NA || Nbl <= 0 || Nt <= 0 || Nf <= 0 || Pstride <= 0 || Pstride % 64 != 0)
        return RIME_EINVAL;
    if (sign != 1 && sign != -1) return RIME_EINVAL;
    AntBwdArgs A{};
    A.antpos = antpos; A.sdir = sdir; A.freqs = freqs; A.gvis = gvis; A.gscale = gscale;
    A.pair_direct = pair_direct; A.pair_conj = pair_conj; A.gpsky = gpsky;
    A.Nant = Nant; A.Nbl = Nbl; A.Nt = Nt; A.Nf = Nf; A.Pstride = Pstride;
    A.st_t = st_t; A.st_f = st_f; A.sign = (double)sign;
    // pixel ranges are independent outputs: split freely for parallelism (>= 256 pixel tiles/block
    // amortise the G staging; fewer when the grid would otherwise be small)
    const int ntile = Pstride / 32;
    int per = 256;
    while (per > 8 && (long)Nt * Nf * ((ntile + per - 1) / per) < 1024) per /= 2;
    A.tiles_per_split = per;
    A.S = (ntile + per - 1) / per;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const size_t lds = 2 * (size_t)MB_GIMG + MF_NA * 3 * sizeof(double);
    dim3 grid(1, Nf, Nt * A.S);
    hipLaunchKernelGGL(fringe_ant_bwd_kernel, grid, dim3(512), lds, st, A);
    return check_launch();
}
