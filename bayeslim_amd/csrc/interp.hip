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

// Workgroups go to the 8 XCDs round-robin in launch order (x fastest), and every XCD has its own L2.  The kernels below
// gather from a beam map (or from the rows of the points around a beam node) that neighbouring tiles share: dealt
// round-robin, every L2 ends up fetching the whole map (measured at C4: 3.4 GB fetched by the fused psky builder for 1.1 GB of
// algorithmic traffic).  xcd_tile() hands XCD k a CONTIGUOUS range of the launch-order tile numbers instead, so tiles that share
// map lines meet in one L2.  Bijective for any grid size; a no-op on a single-XCD partition only in effect, not in result.
// remap == 1: one contiguous range per XCD; remap = C > 1: chunk-cyclic -- of every 8 C consecutive tile numbers XCD k takes
// [k C, (k + 1) C): locality within C tiles, the XCDs' shares interleaved along the tile axis (for tile axes along which
// the work per tile varies); the tail that does not fill 8 C tiles keeps the launch order.
__device__ __forceinline__ void xcd_tile(int remap, int& bx, int& by)
{
    bx = blockIdx.x; by = blockIdx.y;
    if (!remap) return;
    const unsigned gx = gridDim.x, N = gx * gridDim.y;
    const unsigned L = blockIdx.x + blockIdx.y * gx;
    unsigned Lp;
    if (remap == 1) {
        const unsigned k = L & 7u, chunk = N >> 3, rem = N & 7u;
        Lp = k * chunk + (k < rem ? k : rem) + (L >> 3);
    } else {
        const unsigned C = (unsigned)remap, super = 8u * C;
        if (L >= N / super * super) return;
        const unsigned i = L >> 3;
        Lp = (i / C) * super + (L & 7u) * C + i % C;
    }
    bx = (int)(Lp % gx); by = (int)(Lp / gx);
}

// which kernels use it (lab switch RIME_XCD_REMAP = bit mask): 1 psky forward, 2 transposing product of the adjoint,
// 4 sky gradient, 8 CSR scatter.  Measured at C4 (diffuse call, tools/bench_beam_sky.py): forward 0.545 -> 0.489 ms, sky
// gradient - 0.045 ms; the transposing product does not care (+ 0.01) and the CSR scatter is 0.1 ms SLOWER although it
// fetches less (contiguous node ranges are zenith-angle bands with very different list lengths: the XCDs finish at
// different times; it takes the chunk-cyclic form below instead) -- so the default is forward + sky gradient.
enum { XCD_FWD = 1, XCD_BWD = 2, XCD_SKYGRAD = 4, XCD_SCATTER = 8 };
static int xcd_scatter_chunk()
{
    static int v = -1;
    // C4 diffuse adjoint, whole chain: launch order 1.245 ms, one contiguous range per XCD 1.344, chunks of 8 ... 64 1.205, 256
    // 1.238, 512 1.274
    if (v < 0) { const char* e = getenv("RIME_XCD_SCATTER_CHUNK"); v = e ? atoi(e) : 32; }     // lab switch
    return v;
}
static int xcd_remap_enabled(int which)
{
    static int v = -1;
    if (v < 0) { const char* e = getenv("RIME_XCD_REMAP"); v = e ? atoi(e) : (XCD_FWD | XCD_SKYGRAD); }
    return (v & which) ? 1 : 0;
}

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
            // a node of weight 0 contributes nothing even when the map holds a NaN / inf there (RIME's FoV cut is a one-node
            // gather whose padding slots carry weight 0)
#pragma unroll
            for (int k = 0; k < NNN; ++k)
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const T v = row[(size_t)id[k] * NC + c];
                    acc[c] = w[k] != T(0) ? tfma<T>(w[k], v, acc[c]) : acc[c];
                }
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
// list of j four entries at a time (lane = entry slot x 4 row-elements: every contribution is a
// 16-B vector load, 1 KB per wave instruction), the four slots are summed by two shuffles at the
// end -- a fixed order, no atomics.  (The row-major variant read 4 bytes per 64-B line: 11.5 GB
// fetched for 0.8 GB of useful data at C4, 3.3 ms; one entry per wave instruction: 1.19 ms.)
template <typename T> struct svec4 { T x, y, z, w; };

template <typename T>
__global__ void __launch_bounds__(256)
interp_scatter_kernel(const T* __restrict__ goutT, const int* __restrict__ csr_ptr,
                      const int* __restrict__ csr_src, const T* __restrict__ wgts,
                      int RN, int Npb, int Nnn, T* __restrict__ gmT, int remap)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int es = lane >> 4, rq = lane & 15;
    int bx, by;
    xcd_tile(remap, bx, by);
    const int j = bx * 4 + wave;
    if (j >= Npb) return;
    const int e0 = csr_ptr[j], e1 = csr_ptr[j + 1];
    const bool pow2 = (Nnn & (Nnn - 1)) == 0;
    const int sh = __ffs(Nnn) - 1;
    const bool vec_rows = (RN & 3) == 0;
    for (int rb = by * 64; rb < RN; rb += gridDim.y * 64) {
        const int r = rb + 4 * rq;
        T a0 = T(0), a1 = T(0), a2 = T(0), a3 = T(0);
        if (vec_rows) {                                 // uniform (round 5: partial tiles too -- a rank's 32 channels at N = 8 --
                                                        // whose lanes beyond RN read the last valid vector and never store)
            const int rr = min(r, RN - 4);
            // a node's list is a chain entry -> source index -> (weight, row of the source point): the indices of the NEXT
            // four entries of the lane's slot are fetched while the rows of these four are in flight (the list is walked in
            // the same order as by the plain loop below: same sums)
            constexpr int U = 4;
            int sn[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { const int e = e0 + es + 4 * u; sn[u] = e < e1 ? csr_src[e] : -1; }
            for (int eb = e0; eb < e1; eb += 4 * U) {   // uniform
                int sc[U];
#pragma unroll
                for (int u = 0; u < U; ++u) sc[u] = sn[u];
#pragma unroll
                for (int u = 0; u < U; ++u) { const int e = eb + 4 * U + es + 4 * u; sn[u] = e < e1 ? csr_src[e] : -1; }
                T w[U]; svec4<T> v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int src = sc[u] >= 0 ? sc[u] : 0;
                    w[u] = sc[u] >= 0 ? wgts[src] : T(0);
                    v[u] = *reinterpret_cast<const svec4<T>*>(goutT + (size_t)(pow2 ? src >> sh : src / Nnn) * RN + rr);
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (sc[u] >= 0) {
                        a0 = tfma<T>(w[u], v[u].x, a0); a1 = tfma<T>(w[u], v[u].y, a1);
                        a2 = tfma<T>(w[u], v[u].z, a2); a3 = tfma<T>(w[u], v[u].w, a3);
                    }
            }
        } else {
#pragma unroll 2
        for (int e = e0 + es; e < e1; e += 4) {
            const int src = csr_src[e];
            const T w = wgts[src];
            const T* p = goutT + (size_t)(pow2 ? src >> sh : src / Nnn) * RN + r;
            if (vec_rows && r + 3 < RN) {
                const svec4<T> v = *reinterpret_cast<const svec4<T>*>(p);
                a0 = tfma<T>(w, v.x, a0); a1 = tfma<T>(w, v.y, a1); a2 = tfma<T>(w, v.z, a2); a3 = tfma<T>(w, v.w, a3);
            } else {
                if (r < RN) a0 = tfma<T>(w, p[0], a0);
                if (r + 1 < RN) a1 = tfma<T>(w, p[1], a1);
                if (r + 2 < RN) a2 = tfma<T>(w, p[2], a2);
                if (r + 3 < RN) a3 = tfma<T>(w, p[3], a3);
            }
        }
        }
        a0 += __shfl_xor(a0, 16, 64); a1 += __shfl_xor(a1, 16, 64); a2 += __shfl_xor(a2, 16, 64); a3 += __shfl_xor(a3, 16, 64);
        a0 += __shfl_xor(a0, 32, 64); a1 += __shfl_xor(a1, 32, 64); a2 += __shfl_xor(a2, 32, 64); a3 += __shfl_xor(a3, 32, 64);
        if (es == 0) {
            T* o = gmT + (size_t)j * RN + r;
            if (r < RN) o[0] = a0;
            if (r + 1 < RN) o[1] = a1;
            if (r + 2 < RN) o[2] = a2;
            if (r + 3 < RN) o[3] = a3;
        }
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
                       reinterpret_cast<T*>(gmT), xcd_scatter_chunk());
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

// ---------------------------------------------------------------------------------------
// Adjoint of a ONE-NODE gather (the FoV cut of the generic psky path, the redundant inflation: out[r, q] = w[q] m[r, inds[q]])
// on ROW-MAJOR buffers: gm[r, j] = sum over the CSR list of j of w[q] gout[r, q].  Lanes run along the map pixels j: the store is
// coalesced and the reads nearly so (the points that select consecutive pixels are consecutive but for the cut's gaps), so the
// two transposing copies the general kernel needs (goutT, gmT -- and the strided gradient it hands upstream, which the next
// consumer has to make contiguous: 5 ms of a C5 rank step) disappear.  Same fixed summation order: deterministic.
// ---------------------------------------------------------------------------------------
namespace rime {

template <typename T, int EW, int RB>
__global__ void __launch_bounds__(256)
scatter_rows_kernel(const T* __restrict__ gout, long long gstride, const int* __restrict__ csr_ptr, const int* __restrict__ csr_src,
                    const T* __restrict__ wgts, int R, int Npb, T* __restrict__ gm)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int r0 = blockIdx.y * RB;
    if (j >= Npb) return;
    const int e0 = csr_ptr[j], e1 = csr_ptr[j + 1];
    T acc[RB][EW];
#pragma unroll
    for (int k = 0; k < RB; ++k)
#pragma unroll
        for (int c = 0; c < EW; ++c) acc[k][c] = T(0);
    for (int e = e0; e < e1; ++e) {
        const int q = csr_src[e];
        const T w = wgts[q];
#pragma unroll
        for (int k = 0; k < RB; ++k) {
            if (r0 + k < R) {
                const T* p = gout + ((size_t)(r0 + k) * gstride + q) * EW;
#pragma unroll
                for (int c = 0; c < EW; ++c) acc[k][c] = tfma<T>(w, p[c], acc[k][c]);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < RB; ++k) {
        if (r0 + k < R) {
            T* o = gm + ((size_t)(r0 + k) * Npb + j) * EW;
#pragma unroll
            for (int c = 0; c < EW; ++c) o[c] = acc[k][c];
        }
    }
}

} // namespace rime

extern "C" int rime_interp_scatter_rows_bwd(int dtype, int is_complex, const void* gout, long long gout_stride,
                                            const int* csr_ptr, const int* csr_src, const void* wgts, int R, int Npb,
                                            void* gm, void* stream)
{
    // csr_src may be NULL when the index is EMPTY (every weight 0: csr_ptr is all zeros and no entry is read): the result is 0
    if (!gout || !csr_ptr || !wgts || !gm || R <= 0 || Npb <= 0 || gout_stride <= 0) return RIME_EINVAL;
    if (dtype != RIME_F32 && dtype != RIME_F64) return RIME_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    constexpr int RB = 8;
    // rows beyond one grid's reach (65535 x RB) go to further launches on row-offset pointers (was RIME_EUNSUPPORTED)
    const size_t esz = (dtype == RIME_F32 ? 4 : 8) * (is_complex ? 2 : 1);
    const int chunk = 65535 * RB;
    for (int r0 = 0; r0 < R; r0 += chunk) {
        const int Rc = R - r0 < chunk ? R - r0 : chunk;
        dim3 grid((Npb + 255) / 256, (Rc + RB - 1) / RB);
        const char* gi = (const char*)gout + (size_t)r0 * (size_t)gout_stride * esz;
        char* go = (char*)gm + (size_t)r0 * (size_t)Npb * esz;
#define RIME_SR(T, EW) hipLaunchKernelGGL((rime::scatter_rows_kernel<T, EW, RB>), grid, dim3(256), 0, st, (const T*)gi, gout_stride, \
                                          csr_ptr, csr_src, (const T*)wgts, Rc, Npb, (T*)go)
        if (dtype == RIME_F32) { if (is_complex) RIME_SR(float, 2); else RIME_SR(float, 1); }
        else { if (is_complex) RIME_SR(double, 2); else RIME_SR(double, 1); }
#undef RIME_SR
    }
    return rime::check_launch();
}

// ---------------------------------------------------------------------------------------
// Fused psky builder for the 1-pol power-beam case (beam_model.py:238-269 gen_beam's interpolation,
// beam_model.py:1681-1698 cut_sky_fov and the beam x sky product of apply_beam :313-322):
//     psky[r, q] = ( sum_k w[q,k] bmapT[inds[q,k], r] ) * sky[r, cut[q]]      r = channel, q = (t, p)
// instead of three passes (interpolated beam, cut sky, product) over (Nf x Nt x P) tensors.
// cut[q] == Npix marks the zero padding of a time step.  Adjoint:
//     T1[q, r]   = gpsky[r, q] * sky[r, cut[q]]      (transposed: the layout interp_scatter_kernel reads)
//     gsky[r, j] = sum_t gpsky[r, q] * beam_interp[r, q],  q = t Ps + pos[t, j]
//                  (beam re-interpolated, not saved; pos = inverse of cut per time step, -1 = not visible)
// all deterministic (no atomics).
// ---------------------------------------------------------------------------------------
namespace rime {

// Both kernels work on tiles of 64 points x 64 channels in two phases.  Interpolation phase: a lane owns 4
// consecutive channels of one point and the beam map is read NODE-MAJOR (bmapT [Npb][R]), so one vector
// load fetches a node's value for 4 channels and a wave instruction covers 4 points x 64 channels: 4x fewer
// gather instructions than one lane per point walking the channels of a channel-major map (these kernels
// are bound by the address rate of the texture path, not by bytes).  The interpolated tile goes through LDS
// to the product phase, whose lanes run along the points: psky / gpsky / gs rows are contiguous there.
template <typename T> struct vec4 { T x, y, z, w; };

// the NNN node indices and weights of point q; four-node stencils are two 16-byte loads
template <typename T, int NNN>
__device__ __forceinline__ void load_stencil(const int* __restrict__ inds, const T* __restrict__ wgts, size_t q,
                                             int (&id)[NNN], T (&wk)[NNN])
{
    if constexpr (NNN == 4) {
        const int4 i4 = *reinterpret_cast<const int4*>(inds + q * 4);
        id[0] = i4.x; id[1] = i4.y; id[2] = i4.z; id[3] = i4.w;
        if constexpr (sizeof(T) == 4) {
            const vec4<T> w4 = *reinterpret_cast<const vec4<T>*>(wgts + q * 4);
            wk[0] = w4.x; wk[1] = w4.y; wk[2] = w4.z; wk[3] = w4.w;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) wk[k] = wgts[q * 4 + k];
        }
    } else {
#pragma unroll
        for (int k = 0; k < NNN; ++k) { id[k] = inds[q * NNN + k]; wk[k] = wgts[q * NNN + k]; }
    }
}

// points of the tile: q0 + ql (valid when inside Q and not a padding slot of the cut), or qlist[ql] (< 0: none).
// The block's time goes into a chain of dependent memory phases (cut -> stencil -> beam nodes), so everything is loaded
// UNCONDITIONALLY (points without a value read the stencil of point 0 and are zeroed afterwards): no branch stands between the
// loads and the compiler issues the four points' stencils together and then all their node gathers.
template <typename T, int NNN>
__device__ __forceinline__ void interp_tile(const T* __restrict__ bmapT, const int* __restrict__ inds,
                                            const T* __restrict__ wgts, const int* __restrict__ cut, int R, int Npix,
                                            int Q, int Nnn, int q0, int r0, T (*tile)[65], const int* qlist = nullptr)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int qs = lane >> 4, rq = lane & 15;
    const int r = r0 + 4 * rq;
    const bool vec_ok = (R & 3) == 0 && r + 3 < R;
    if constexpr (NNN > 0) {
        if (vec_ok) {
            int qc[4]; bool ok[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ql = 16 * w + 4 * j + qs, q = qlist ? qlist[ql] : q0 + ql;
                ok[j] = qlist ? q >= 0 : (q < Q && cut[min(q, Q - 1)] < Npix);
                qc[j] = ok[j] ? q : 0;
            }
            int id[4][NNN]; T wk[4][NNN];
#pragma unroll
            for (int j = 0; j < 4; ++j) load_stencil<T, NNN>(inds, wgts, (size_t)qc[j], id[j], wk[j]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                T b0 = T(0), b1 = T(0), b2 = T(0), b3 = T(0);
#pragma unroll
                for (int k = 0; k < NNN; ++k) {
                    const vec4<T> v = *reinterpret_cast<const vec4<T>*>(bmapT + (size_t)id[j][k] * R + r);
                    b0 = tfma<T>(wk[j][k], v.x, b0); b1 = tfma<T>(wk[j][k], v.y, b1);
                    b2 = tfma<T>(wk[j][k], v.z, b2); b3 = tfma<T>(wk[j][k], v.w, b3);
                }
                const int ql = 16 * w + 4 * j + qs;
                tile[ql][4 * rq] = ok[j] ? b0 : T(0); tile[ql][4 * rq + 1] = ok[j] ? b1 : T(0);
                tile[ql][4 * rq + 2] = ok[j] ? b2 : T(0); tile[ql][4 * rq + 3] = ok[j] ? b3 : T(0);
            }
            return;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ql = 16 * w + 4 * j + qs, q = qlist ? qlist[ql] : q0 + ql;
        T b0 = T(0), b1 = T(0), b2 = T(0), b3 = T(0);
        if (qlist ? q >= 0 : (q < Q && cut[q] < Npix)) {
            const int nn = NNN > 0 ? NNN : Nnn;
            for (int k = 0; k < nn; ++k) {
                const T wk = wgts[(size_t)q * nn + k];
                const T* src = bmapT + (size_t)inds[(size_t)q * nn + k] * R + r;
                if (vec_ok) {
                    const vec4<T> v = *reinterpret_cast<const vec4<T>*>(src);
                    b0 = tfma<T>(wk, v.x, b0); b1 = tfma<T>(wk, v.y, b1); b2 = tfma<T>(wk, v.z, b2); b3 = tfma<T>(wk, v.w, b3);
                } else {
                    if (r < R) b0 = tfma<T>(wk, src[0], b0);
                    if (r + 1 < R) b1 = tfma<T>(wk, src[1], b1);
                    if (r + 2 < R) b2 = tfma<T>(wk, src[2], b2);
                    if (r + 3 < R) b3 = tfma<T>(wk, src[3], b3);
                }
            }
        }
        tile[ql][4 * rq] = b0; tile[ql][4 * rq + 1] = b1; tile[ql][4 * rq + 2] = b2; tile[ql][4 * rq + 3] = b3;
    }
}

#ifndef RIME_BSFWD_BLOCKS
#define RIME_BSFWD_BLOCKS 8
#endif
template <typename T, int NNN>
__global__ void __launch_bounds__(256, (NNN > 0 && NNN <= 4 && sizeof(T) == 4) ? RIME_BSFWD_BLOCKS : 1)
beam_sky_fwd_kernel(const T* __restrict__ bmapT, const T* __restrict__ sky, const int* __restrict__ inds,
                    const T* __restrict__ wgts, const int* __restrict__ cut, int R, int Npb, int Npix,
                    int Q, int Nnn, T* __restrict__ out, int remap)
{
    __shared__ T tile[64][65];
    int bx, by;
    xcd_tile(remap, bx, by);
    const int q0 = bx * 64, r0 = by * 64;
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    const int q = q0 + lx;
    // the sky values of the product phase are fetched first: their latency runs under the interpolation phase's
    const int c = q < Q ? cut[q] : Npix;
    T sv[16];
#pragma unroll
    for (int k16 = 0; k16 < 16; ++k16) {
        const int r = r0 + ly + 4 * k16;
        sv[k16] = (c < Npix && r < R) ? sky[(size_t)r * Npix + c] : T(0);
    }
    interp_tile<T, NNN>(bmapT, inds, wgts, cut, R, Npix, Q, Nnn, q0, r0, tile);
    __syncthreads();
    if (q >= Q) return;
#pragma unroll
    for (int k16 = 0; k16 < 16; ++k16) {
        const int rl = ly + 4 * k16, r = r0 + rl;
        if (r < R) out[(size_t)r * Q + q] = c < Npix ? tile[lx][rl] * sv[k16] : T(0);
    }
}

// adjoint, beam side: T1[q, r] = gpsky[r, q] * sky[r, cut[q]] -- read along q, written along r through LDS
template <typename T>
__global__ void __launch_bounds__(256)
beam_sky_bwd_kernel(const T* __restrict__ gps, const T* __restrict__ sky, const int* __restrict__ cut,
                    int R, int Npix, int Q, T* __restrict__ T1, int remap)
{
    __shared__ T tile[64][65];
    int bx, by;
    xcd_tile(remap, bx, by);                           // neighbouring point tiles share the sky lines their cuts straddle
    const int q0 = bx * 64, r0 = by * 64;
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    const int q = q0 + lx;
    const int c = q < Q ? cut[q] : Npix;
#pragma unroll 4
    for (int k16 = 0; k16 < 16; ++k16) {
        const int rl = ly + 4 * k16, r = r0 + rl;
        tile[rl][lx] = (c < Npix && r < R) ? gps[(size_t)r * Q + q] * sky[(size_t)r * Npix + c] : T(0);
    }
    __syncthreads();
#pragma unroll 4
    for (int k16 = 0; k16 < 16; ++k16) {
        const int ql = ly + 4 * k16, qq = q0 + ql, r = r0 + lx;
        if (qq < Q && r < R) T1[(size_t)qq * R + r] = tile[lx][ql];
    }
}

// adjoint, sky side: gsky[r, j] = sum_t gpsky[r, q] * beam_interp[r, q], q = t Ps + pos[t, j] -- a tile of 64 sky
// pixels x 64 channels walks the time steps, re-interpolating the beam for the pixels visible at t (the
// (Nf x Nt P) product gpsky * beam is never written: it would be one more write and one more read of the largest
// tensor of the step)
#ifndef RIME_SKYGRAD_BLOCKS
#define RIME_SKYGRAD_BLOCKS 4
#endif
template <typename T, int NNN>
__global__ void __launch_bounds__(256, (NNN > 0 && NNN <= 4 && sizeof(T) == 4) ? RIME_SKYGRAD_BLOCKS : 1)
sky_grad_kernel(const T* __restrict__ gps, const T* __restrict__ bmapT, const int* __restrict__ inds,
                const T* __restrict__ wgts, const int* __restrict__ pos, int R, int Npix, int Nt, int Ps, int Nnn,
                T* __restrict__ gsky, int remap, int Tc)
{
    // blockIdx.z = time split: steps [z Tc, min(Nt, (z + 1) Tc)) into plane z of gsky (a workspace when there are several: workloads
    // of many time steps and few sky pixels -- C2: 192 tiles for 256 CUs, each walking 30 steps -- are split over time and the
    // planes summed in a fixed order by plane_sum_kernel)
    const int tz0 = blockIdx.z * Tc;
    const size_t Q = (size_t)Nt * Ps;                  // row length of gpsky: ALL time steps
    Nt = min(Nt, tz0 + Tc);
    gsky += (size_t)blockIdx.z * R * Npix;
    __shared__ T tile[64][65];
    __shared__ int qsel[64];
    int bx, by;
    xcd_tile(remap, bx, by);
    const int j0 = bx * 64, r0 = by * 64;
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    const int j = j0 + lx;
    T acc[16];
#pragma unroll
    for (int k16 = 0; k16 < 16; ++k16) acc[k16] = T(0);
    if constexpr (NNN == 4) {
        if ((R & 3) == 0) {                            // uniform: every lane's 4 channels are one vector load (round 5: tiles that
                                                       // reach past the last channel too -- their lanes beyond R are masked; a rank's
                                                       // share of 32 channels at N = 8 took the slow path below)
            // Software pipeline over the time steps: the chain pos -> stencil -> beam nodes of step t+1 is walked while step
            // t is contracted.  The stencils of the tile's 64 pixels (256 indices + 256 weights: one of each per thread) go
            // through LDS, double buffered: thread (pixel pl, node k) loads its entry for step t+1 during step t (from the
            // position it loaded during t-1) and stores it before the step's barrier; the interpolation phase reads
            // its four pixels' stencils with two ds_read_b128 each.  Only the node gathers + the gpsky column of the step
            // itself are waited for, and a lane carries 3 registers of pipeline state.  (Two other forms were slower at
            // C4: the stencils prefetched into registers -- 220 registers, two blocks per CU, 32 scalar stencil loads per
            // lane and step: 0.56 ms; a wave-autonomous form without LDS and barriers, each lane interpolating exactly the
            // 4 pixels x 4 channels whose gpsky values it multiplies: 1.17 ms, its gpsky loads touch 16 rows per
            // instruction instead of one.)
            __shared__ int sten_i[2][64][4];
            __shared__ T sten_w[2][64][4];
            __shared__ int sten_q[2][64];
            const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
            const int qs = lane >> 4, rq = lane & 15;
            const int r = min(r0 + 4 * rq, R - 4);               // (clamped: lanes past the last channel read row R - 4 and store zeros)
            const bool rin = r0 + 4 * rq < R;
            const int pl = threadIdx.x >> 2, pk = threadIdx.x & 3, jp = j0 + pl;
            auto q_of = [&](int t) {
                const int pp = (t < Nt && jp < Npix) ? pos[(size_t)t * Npix + jp] : -1;
                return pp >= 0 ? t * Ps + pp : -1;
            };
            {
                const int q0 = q_of(tz0);
                const size_t qc = q0 >= 0 ? (size_t)q0 : 0;
                sten_i[0][pl][pk] = inds[qc * 4 + pk];
                sten_w[0][pl][pk] = wgts[qc * 4 + pk];
                if (pk == 0) sten_q[0][pl] = q0;
            }
            int q1 = q_of(tz0 + 1);
            int pprod = (j < Npix && tz0 < Nt) ? pos[(size_t)tz0 * Npix + j] : -1;
            for (int t = tz0; t < Nt; ++t) {
                const int q = pprod >= 0 ? t * Ps + pprod : -1;
                if (t + 1 < Nt) pprod = j < Npix ? pos[(size_t)(t + 1) * Npix + j] : -1;
                const size_t qc1 = q1 >= 0 ? (size_t)q1 : 0;
                const int nid = inds[qc1 * 4 + pk];
                const T nw = wgts[qc1 * 4 + pk];
                const int q2 = q_of(t + 2);
                const int cur = (t - tz0) & 1;
                const bool any = __syncthreads_or(q >= 0);       // some pixel of the tile is above the horizon at t (uniform)
                T g[16];
                if (any) {
#pragma unroll
                    for (int k16 = 0; k16 < 16; ++k16)
                        g[k16] = (q >= 0 && r0 + ly + 4 * k16 < R) ? gps[(size_t)(r0 + ly + 4 * k16) * Q + q] : T(0);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int ql = 16 * w + 4 * u + qs;
                        const int4 id = *reinterpret_cast<const int4*>(&sten_i[cur][ql][0]);
                        const vec4<T> wk = *reinterpret_cast<const vec4<T>*>(&sten_w[cur][ql][0]);
                        const bool ok = sten_q[cur][ql] >= 0 && rin;
                        const vec4<T> v0 = *reinterpret_cast<const vec4<T>*>(bmapT + (size_t)id.x * R + r);
                        const vec4<T> v1 = *reinterpret_cast<const vec4<T>*>(bmapT + (size_t)id.y * R + r);
                        const vec4<T> v2 = *reinterpret_cast<const vec4<T>*>(bmapT + (size_t)id.z * R + r);
                        const vec4<T> v3 = *reinterpret_cast<const vec4<T>*>(bmapT + (size_t)id.w * R + r);
                        T b0 = tfma<T>(wk.x, v0.x, T(0)), b1 = tfma<T>(wk.x, v0.y, T(0)), b2 = tfma<T>(wk.x, v0.z, T(0)), b3 = tfma<T>(wk.x, v0.w, T(0));
                        b0 = tfma<T>(wk.y, v1.x, b0); b1 = tfma<T>(wk.y, v1.y, b1); b2 = tfma<T>(wk.y, v1.z, b2); b3 = tfma<T>(wk.y, v1.w, b3);
                        b0 = tfma<T>(wk.z, v2.x, b0); b1 = tfma<T>(wk.z, v2.y, b1); b2 = tfma<T>(wk.z, v2.z, b2); b3 = tfma<T>(wk.z, v2.w, b3);
                        b0 = tfma<T>(wk.w, v3.x, b0); b1 = tfma<T>(wk.w, v3.y, b1); b2 = tfma<T>(wk.w, v3.z, b2); b3 = tfma<T>(wk.w, v3.w, b3);
                        tile[ql][4 * rq] = ok ? b0 : T(0); tile[ql][4 * rq + 1] = ok ? b1 : T(0);
                        tile[ql][4 * rq + 2] = ok ? b2 : T(0); tile[ql][4 * rq + 3] = ok ? b3 : T(0);
                    }
                }
                sten_i[cur ^ 1][pl][pk] = nid;
                sten_w[cur ^ 1][pl][pk] = nw;
                if (pk == 0) sten_q[cur ^ 1][pl] = q1;
                q1 = q2;
                if (any) {
                    __syncthreads();
#pragma unroll
                    for (int k16 = 0; k16 < 16; ++k16) acc[k16] = tfma<T>(g[k16], tile[lx][ly + 4 * k16], acc[k16]);
                }
            }
            if (j < Npix) {
#pragma unroll
                for (int k16 = 0; k16 < 16; ++k16)
                    if (r0 + ly + 4 * k16 < R) gsky[(size_t)(r0 + ly + 4 * k16) * Npix + j] = acc[k16];
            }
            return;
        }
    }
    // the block is a chain of dependent memory phases per time step (pos -> stencil -> beam nodes -> gpsky);
    // pos of the next step and the gpsky column of this one are fetched ahead of the interpolation phase
    int pnext = (j < Npix && tz0 < Nt) ? pos[(size_t)tz0 * Npix + j] : -1;
    for (int t = tz0; t < Nt; ++t) {
        const int q = pnext >= 0 ? t * Ps + pnext : -1;
        if (threadIdx.x < 64) qsel[threadIdx.x] = q;
        if (t + 1 < Nt) pnext = j < Npix ? pos[(size_t)(t + 1) * Npix + j] : -1;
        if (!__syncthreads_or(q >= 0)) continue;         // none of the 64 pixels is above the horizon at t (uniform)
        T g[16];
#pragma unroll
        for (int k16 = 0; k16 < 16; ++k16) {
            const int r = r0 + ly + 4 * k16;
            g[k16] = (q >= 0 && r < R) ? gps[(size_t)r * Q + q] : T(0);
        }
        interp_tile<T, NNN>(bmapT, inds, wgts, nullptr, R, Npix, 0, Nnn, 0, r0, tile, qsel);
        __syncthreads();
#pragma unroll
        for (int k16 = 0; k16 < 16; ++k16) acc[k16] = tfma<T>(g[k16], tile[lx][ly + 4 * k16], acc[k16]);
        __syncthreads();
    }
    if (j < Npix) {
#pragma unroll
        for (int k16 = 0; k16 < 16; ++k16) {
            const int r = r0 + ly + 4 * k16;
            if (r < R) gsky[(size_t)r * Npix + j] = acc[k16];
        }
    }
}

// sums the S time-split planes of the sky gradient in plane order (fixed: bitwise reproducible)
template <typename T>
__global__ void __launch_bounds__(256)
plane_sum_kernel(const T* __restrict__ part, T* __restrict__ out, size_t len, int S)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < len; i += (size_t)gridDim.x * 256) {
        T v = part[i];
        for (int z = 1; z < S; ++z) v += part[(size_t)z * len + i];
        out[i] = v;
    }
}

// time splits of the sky gradient: enough blocks for ~8 per CU, at least 4 time steps per split
static int sky_grad_splits(int R, int Npix, int Nt)
{
    const long blocks = (long)((Npix + 63) / 64) * ((R + 63) / 64);
    long S = std::min<long>((Nt + 3) / 4, (2048 + blocks - 1) / blocks);
    if (const char* e = getenv("RIME_SKYGRAD_SPLITS")) { const long v = atol(e); if (v >= 1 && v <= Nt) S = v; }     // lab
    return (int)std::max<long>(1, std::min<long>(S, 64));
}

template <typename T>
static int beam_sky_fwd_launch(const void* bmap, const void* sky, const int* inds, const void* wgts, const int* cut,
                               int R, int Npb, int Npix, int Q, int Nnn, void* out, hipStream_t st)
{
    dim3 grid((Q + 63) / 64, (R + 63) / 64), block(256);
    const T* b_ = reinterpret_cast<const T*>(bmap); const T* s_ = reinterpret_cast<const T*>(sky);
    const T* w_ = reinterpret_cast<const T*>(wgts); T* o_ = reinterpret_cast<T*>(out);
    const int remap = xcd_remap_enabled(XCD_FWD);
#define RIME_BS(N) hipLaunchKernelGGL((beam_sky_fwd_kernel<T, N>), grid, block, 0, st, b_, s_, inds, w_, cut, R, Npb, Npix, Q, Nnn, o_, remap)
    switch (Nnn) {
        case 1: RIME_BS(1); break;
        case 4: RIME_BS(4); break;
        case 9: RIME_BS(9); break;
        case 16: RIME_BS(16); break;
        default: RIME_BS(0); break;
    }
#undef RIME_BS
    return check_launch();
}

} // namespace rime

extern "C" int rime_beam_sky_fwd(int dtype, const void* bmap, const void* sky, const int* inds, const void* wgts,
                                 const int* cut, int R, int Npb, int Npix, int Q, int Nnn, void* psky, void* stream)
{
    if (!bmap || !sky || !inds || !wgts || !cut || !psky) return RIME_EINVAL;
    if (R <= 0 || Npb <= 0 || Npix <= 0 || Q <= 0 || Nnn <= 0) return RIME_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == RIME_F32) return beam_sky_fwd_launch<float>(bmap, sky, inds, wgts, cut, R, Npb, Npix, Q, Nnn, psky, st);
    if (dtype == RIME_F64) return beam_sky_fwd_launch<double>(bmap, sky, inds, wgts, cut, R, Npb, Npix, Q, Nnn, psky, st);
    return RIME_EINVAL;
}

extern "C" size_t rime_beam_sky_bwd_workspace(int dtype, int R, int Npix, int Nt)
{
    if ((dtype != RIME_F32 && dtype != RIME_F64) || R <= 0 || Npix <= 0 || Nt <= 0) return 0;
    const int S = sky_grad_splits(R, Npix, Nt);
    return S > 1 ? (size_t)S * R * Npix * (dtype == RIME_F64 ? 8 : 4) : 0;
}

extern "C" int rime_beam_sky_bwd(int dtype, const void* gpsky, const void* bmapT, const void* sky, const int* inds,
                                 const void* wgts, const int* cut, const int* pos, int R, int Npb, int Npix,
                                 int Nt, int Ps, int Nnn, void* T1, void* gsky, void* workspace, size_t workspace_bytes,
                                 void* stream)
{
    if (!gpsky || !bmapT || !sky || !inds || !wgts || !cut || !pos || !T1 || !gsky) return RIME_EINVAL;
    if (R <= 0 || Npb <= 0 || Npix <= 0 || Nt <= 0 || Ps <= 0 || Nnn <= 0) return RIME_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int Q = Nt * Ps;
    const int remap = xcd_remap_enabled(XCD_SKYGRAD), remap_b = xcd_remap_enabled(XCD_BWD);
    // time splits of the sky gradient (their partial planes live in the workspace; without one the steps are walked in one go)
    int S = sky_grad_splits(R, Npix, Nt);
    if (S > 1 && (!workspace || workspace_bytes < rime_beam_sky_bwd_workspace(dtype, R, Npix, Nt))) {
        if (workspace) return RIME_EWORKSPACE;
        S = 1;
    }
    const int Tc = (Nt + S - 1) / S;
    S = (Nt + Tc - 1) / Tc;
    void* sg_out = S > 1 ? workspace : gsky;
    dim3 g1((Q + 63) / 64, (R + 63) / 64), g2((Npix + 63) / 64, (R + 63) / 64, S);
#define RIME_SG(TT, N) hipLaunchKernelGGL((sky_grad_kernel<TT, N>), g2, dim3(256), 0, st, (const TT*)gpsky, (const TT*)bmapT, \
        inds, (const TT*)wgts, pos, R, Npix, Nt, Ps, Nnn, (TT*)sg_out, remap, Tc)
#define RIME_SG_ALL(TT) switch (Nnn) { case 1: RIME_SG(TT, 1); break; case 4: RIME_SG(TT, 4); break; \
        case 9: RIME_SG(TT, 9); break; case 16: RIME_SG(TT, 16); break; default: RIME_SG(TT, 0); break; }
    const size_t len = (size_t)R * Npix;
    const int nb = (int)std::min<size_t>((len + 255) / 256, 4096);
    if (dtype == RIME_F32) {
        hipLaunchKernelGGL((beam_sky_bwd_kernel<float>), g1, dim3(256), 0, st, (const float*)gpsky, (const float*)sky, cut,
                           R, Npix, Q, (float*)T1, remap_b);
        RIME_SG_ALL(float)
        if (S > 1) hipLaunchKernelGGL((plane_sum_kernel<float>), dim3(nb), dim3(256), 0, st, (const float*)workspace, (float*)gsky, len, S);
    } else if (dtype == RIME_F64) {
        hipLaunchKernelGGL((beam_sky_bwd_kernel<double>), g1, dim3(256), 0, st, (const double*)gpsky, (const double*)sky, cut,
                           R, Npix, Q, (double*)T1, remap_b);
        RIME_SG_ALL(double)
        if (S > 1) hipLaunchKernelGGL((plane_sum_kernel<double>), dim3(nb), dim3(256), 0, st, (const double*)workspace, (double*)gsky, len, S);
    } else return RIME_EINVAL;
#undef RIME_SG_ALL
#undef RIME_SG
    return check_launch();
}
