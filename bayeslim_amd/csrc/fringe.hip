// fringe.hip -- fused fringe-sum kernels for gfx950 (MI355X), forward and backward.
//
//   vis[pp,b,t,f]     = sum_p   psky[t,mp(b),pp,f,p] * exp(sign 2 pi i nu_f/c  b . s_{t,p})
//   gpsky[t,mp,pp,f,p] = sum_b  conj(F[b,t,f,p]) * gvis[pp,b,t,f]
//
// Replaces ArrayModel.gen_fringe (telescope_model.py:310-358) + the product/sum of
// RIME._prod_and_sum (rime_model.py:423-429) and its autograd backward.  The (Nbl,Nf,P)
// fringe tensor never exists: every wavefront regenerates the fringe in registers.
//
// Design (CDNA4):
//   * forward: one lane = one baseline x one chunk of CH channels; the lane keeps 2*Npp*CH
//     accumulators in VGPRs and walks over ALL pixels, so no cross-lane reduction is needed.
//     Pixel tiles (direction cosines in f64 + the psky chunk) are staged in LDS and read
//     back as wave-uniform (broadcast, conflict-free) ds_read_b128.
//   * backward: transposed -- one lane = PIX pixels x one chunk, walking over baselines whose
//     vectors and gvis chunk are staged in LDS.  Deterministic: no atomics anywhere.
//   * per (baseline,pixel): geometric delay and anchor phase in f64 (phases reach 1e3 turns;
//     f32 would lose 1e-4 rad), reduced to [-1/2,1/2] turns, then ONE f32 sincos for the
//     anchor channel and one for the per-channel step; the other channels follow by rotation,
//     half the chunk upwards and half downwards from the central anchor (two independent
//     dependency chains, <= CH/2 steps of roundoff growth).
//   * rotation by 3 shears (lifting: x+=a*y; y+=b*x; x+=a*y with a=-tan(th/2), b=sin th) =
//     3 FMAs instead of 2 MUL + 2 FMA, used when the host can bound |step| < 0.3 turn.
//   * long sums are flushed to memory every 2048 terms so f32 accumulation error stays at
//     eps*sqrt(2048) instead of eps*sqrt(P).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rime_common.h"

namespace rime {

constexpr int TP = 64;            // pixels per LDS tile (forward)
constexpr int TB = 64;            // baselines per LDS tile (backward)
constexpr int FLUSH_TILES = 32;   // flush accumulators every 32 tiles = 2048 terms

// MODE_*_NU ("near uniform"): channel centres deviate from a uniform grid by tiny residuals
// (e.g. a float32-rounded linspace: +-8 Hz at 180 MHz).  The rotation recurrence runs on the
// fitted uniform grid and each channel gets the first-order phase correction
// (x, y) -> (x - phi y, y + phi x), phi = 2 pi tau eps_k / c; the caller guarantees |phi| < 2e-3
// so the neglected phi^2/2 stays below 2e-6.
enum { MODE_DIRECT = 0, MODE_ROT = 1, MODE_LIFT = 2, MODE_ROT_NU = 3, MODE_LIFT_NU = 4 };
constexpr bool mode_is_lift(int m) { return m == MODE_LIFT || m == MODE_LIFT_NU; }
constexpr bool mode_is_nu(int m) { return m == MODE_ROT_NU || m == MODE_LIFT_NU; }

struct FringeArgs {
    const double* blvecs;
    const double* sdir;
    const double* freqs;
    const void* in;
    void* out;
    void* ws;
    const int* bl_order;
    int bl_off, bl_cnt, mp;
    int Nbl, Nt, Nf, Pstride, Nmp;
    int S, tiles_per_split;
    int nbx;                   // blocks per (time, split): blockIdx.x = (t * S + split) * nbx + bx (grid.z is capped at 65535)
    long long st_t, st_mp, st_pp, st_f;   // psky / gpsky strides [elements] of (time, model pair, pol product, channel)
    double sign;
    double freq0_c, dfreq_c;      // freq0 / c, dfreq / c   [turns per metre]
};

// ---------------------------------------------------------------------------------------
// sin / cos of an angle given in TURNS, |r| <= 1/2: the hardware v_sin_f32 / v_cos_f32 take
// their argument in revolutions; measured max abs error on gfx950 over [-1/2, 1/2] is 1.24e-7
// (profiles/r01_microbench_gfx950.txt) at ~3.2 FMA issue slots each -- versus ~20 slots for a
// quadrant-reduced polynomial of the same accuracy.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void sincos_turns(float r, float& s, float& c)
{
    s = __builtin_amdgcn_sinf(r);
    c = __builtin_amdgcn_cosf(r);
}

__device__ __forceinline__ void sincos_turns(double r, double& s, double& c)
{
    sincospi(2.0 * r, &s, &c);
}

// phase (turns, f64) -> reduced T in [-1/2, 1/2]
template <typename T>
__device__ __forceinline__ T reduce_turns(double ph)
{
    return (T)(ph - rint(ph));
}

// One (baseline, pixel) pair: generate the CH fringe values of the chunk and hand each to
// `sink(k, x, y)` (x = Re F, y = Im F).  `tau` = sign * b.s [m]; nu_c = anchor freq / c;
// dnu = channel spacing / c; fk_c = per-channel freq / c table (MODE_DIRECT only).
template <typename T, int CH, int MODE, bool NOROT = false, typename Sink0>
__device__ __forceinline__ void fringe_chunk(double tau, double nu_c, double dnu,
                                             const double* __restrict__ fk_c, Sink0&& sink0)
{
    constexpr int KC = CH / 2;
    // near-uniform grids: first-order per-channel phase correction in front of the consumer
    const T* e_tab = reinterpret_cast<const T*>(fk_c);
    const T tauf = (T)tau;
    auto sink = [&](int k, T x, T y) {
        if constexpr (mode_is_nu(MODE)) {
            const T ph = tauf * e_tab[k];
            sink0(k, tfma<T>(-ph, y, x), tfma<T>(ph, x, y));
        } else {
            sink0(k, x, y);
        }
    };
    if constexpr (NOROT) {       // lab ablation: anchor only, every channel gets the same phasor
        T zs, zc;
        sincos_turns(reduce_turns<T>(tau * nu_c), zs, zc);
#pragma unroll
        for (int k = 0; k < CH; ++k) sink(k, zc, zs);
        return;
    }
    if constexpr (MODE == MODE_DIRECT) {
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            T s, c;
            sincos_turns(reduce_turns<T>(tau * fk_c[k]), s, c);
            sink(k, c, s);
        }
    } else {
        T zs, zc, ws, wc;
        sincos_turns(reduce_turns<T>(tau * nu_c), zs, zc);
        if constexpr (mode_is_lift(MODE)) {
            // the host guarantees |step| < 0.3 turn here: no range reduction needed
            sincos_turns((T)(tau * dnu), ws, wc);
        } else {
            sincos_turns(reduce_turns<T>(tau * dnu), ws, wc);
        }
        sink(KC, zc, zs);
        if constexpr (mode_is_lift(MODE)) {
            // rotation by +-theta as three shears
            T a = -ws * __builtin_amdgcn_rcpf(T(1) + wc);
            T x = zc, y = zs;
#pragma unroll
            for (int k = KC + 1; k < CH; ++k) {
                x = fmaf(a, y, x);
                y = fmaf(ws, x, y);
                x = fmaf(a, y, x);
                sink(k, x, y);
            }
            x = zc; y = zs;
#pragma unroll
            for (int k = KC - 1; k >= 0; --k) {
                x = fmaf(-a, y, x);
                y = fmaf(-ws, x, y);
                x = fmaf(-a, y, x);
                sink(k, x, y);
            }
        } else {
            T x = zc, y = zs;
#pragma unroll
            for (int k = KC + 1; k < CH; ++k) {
                T xn = x * wc - y * ws;
                T yn = x * ws + y * wc;
                x = xn; y = yn;
                sink(k, x, y);
            }
            x = zc; y = zs;
#pragma unroll
            for (int k = KC - 1; k >= 0; --k) {
                T xn = x * wc + y * ws;
                T yn = y * wc - x * ws;
                x = xn; y = yn;
                sink(k, x, y);
            }
        }
    }
}

template <typename T, int NPP, bool CPLX, int CH>
struct Geom {
    static constexpr int NC = CPLX ? 2 : 1;
    static constexpr int ROW = NPP * CH * NC;                    // T elements per pixel / baseline row
    static constexpr int PAD = 16 / sizeof(T);                   // keep rows 16-B aligned
    static constexpr int ASTRIDE = ROW + PAD;                    // forward LDS row stride
    static constexpr int GROW = NPP * CH * 2;                    // backward: gvis chunk per baseline
    static constexpr int GSTRIDE = GROW + PAD;
};

// ---------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------
// ABL (tools/fringe_lab.hip only; the library always instantiates ABL = 0): timing ablations
//   1: no LDS reads of psky (lane-constant instead)   2: delay tau from f32 (no f64 math)
//   4: skip the accumulate FMAs                        8: skip the rotation chain
template <typename T, int NPP, bool CPLX, int CH, int MODE, int ABL = 0, int WPS = 1>
__global__ void __launch_bounds__(256, WPS)
fringe_fwd_kernel(FringeArgs A)
{
    using G = Geom<T, NPP, CPLX, CH>;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    double* s_lds = reinterpret_cast<double*>(smem_raw);                        // [3][TP]
    double* f_lds = s_lds + 3 * TP;                                             // [CH] (direct mode)
    T* a_lds = reinterpret_cast<T*>(f_lds + CH);                                // [TP][ASTRIDE]

    const int tid = threadIdx.x;
    const int bxi = blockIdx.x % A.nbx, tsi = blockIdx.x / A.nbx;
    const int slot = bxi * blockDim.x + tid;
    const bool active = slot < A.bl_cnt;
    const int b = active ? (A.bl_order ? A.bl_order[A.bl_off + slot] : A.bl_off + slot) : 0;
    const int k0 = blockIdx.y * CH;
    const int t = tsi / A.S;
    const int split = tsi % A.S;
    const int nk = min(CH, A.Nf - k0);

    const double bx = A.sign * A.blvecs[3 * b + 0];
    const double by = A.sign * A.blvecs[3 * b + 1];
    const double bz = A.sign * A.blvecs[3 * b + 2];
    const double nu_c = A.freq0_c + (double)(k0 + CH / 2) * A.dfreq_c;
    const double dnu = A.dfreq_c;

    if constexpr (MODE == MODE_DIRECT) {
        for (int i = tid; i < CH; i += blockDim.x)
            f_lds[i] = (i < nk) ? A.freqs[k0 + i] * (1.0 / 2.99792458e8) : 0.0;
    } else if constexpr (mode_is_nu(MODE)) {
        T* e_lds = reinterpret_cast<T*>(f_lds);
        for (int i = tid; i < CH; i += blockDim.x)
            e_lds[i] = (i < nk) ? (T)(6.283185307179586 * (A.freqs[k0 + i] * (1.0 / 2.99792458e8)
                                      - (A.freq0_c + (double)(k0 + i) * A.dfreq_c))) : T(0);
    }

    T accr[NPP][CH], acci[NPP][CH];
#pragma unroll
    for (int q = 0; q < NPP; ++q)
#pragma unroll
        for (int k = 0; k < CH; ++k) { accr[q][k] = T(0); acci[q][k] = T(0); }

    const int ntiles = A.Pstride / TP;
    const int tile_begin = split * A.tiles_per_split;
    const int tile_end = min(ntiles, tile_begin + A.tiles_per_split);

    const T* psky = reinterpret_cast<const T*>(A.in) + ((size_t)t * A.st_t + (size_t)A.mp * A.st_mp) * G::NC;
    const double* sd = A.sdir + (size_t)t * 3 * A.Pstride;

    // destination: vis itself, or this split's partial slab in the workspace
    const size_t vis_elems = (size_t)NPP * A.Nbl * A.Nt * A.Nf * 2;
    T* dst = (A.S == 1) ? reinterpret_cast<T*>(A.out)
                        : reinterpret_cast<T*>(A.ws) + (size_t)split * vis_elems;
    bool first = true;
    auto flush = [&]() {
        if (active) {
#pragma unroll
            for (int q = 0; q < NPP; ++q) {
                T* o = dst + ((((size_t)q * A.Nbl + b) * A.Nt + t) * A.Nf + k0) * 2;
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    if (k < nk) {
                        T re = accr[q][k], im = acci[q][k];
                        if (!first) { re += o[2 * k]; im += o[2 * k + 1]; }
                        o[2 * k] = re; o[2 * k + 1] = im;
                    }
                    accr[q][k] = T(0); acci[q][k] = T(0);
                }
            }
        }
        first = false;
    };

    int since_flush = 0;
    for (int tile = tile_begin; tile < tile_end; ++tile) {
        __syncthreads();
        const int p0 = tile * TP;
        for (int i = tid; i < 3 * TP; i += blockDim.x)
            s_lds[i] = sd[(size_t)(i / TP) * A.Pstride + p0 + (i % TP)];
        // psky chunk: global [q][f][p][c] -> LDS [p][q][k][c]
        constexpr int NEL = NPP * CH * TP * G::NC;
        for (int i = tid; i < NEL; i += blockDim.x) {
            int c = i % G::NC;
            int pp = (i / G::NC) % TP;
            int k = (i / (G::NC * TP)) % CH;
            int q = i / (G::NC * TP * CH);
            T v = T(0);
            if (k < nk)
                v = psky[((size_t)q * A.st_pp + (size_t)(k0 + k) * A.st_f + p0 + pp) * G::NC + c];
            a_lds[pp * G::ASTRIDE + (q * CH + k) * G::NC + c] = v;
        }
        __syncthreads();

        if (active) {
            for (int pp = 0; pp < TP; ++pp) {
                double tau;
                if constexpr (ABL & 2)
                    tau = (double)((float)bx * (float)s_lds[pp] + (float)by * (float)s_lds[TP + pp]);
                else
                    tau = bx * s_lds[pp] + by * s_lds[TP + pp] + bz * s_lds[2 * TP + pp];
                const T* arow = (ABL & 1) ? a_lds + (tid & 3) * G::ASTRIDE : a_lds + pp * G::ASTRIDE;
                fringe_chunk<T, CH, MODE, (ABL & 8) != 0>(tau, nu_c, dnu, f_lds, [&](int k, T x, T y) {
                    if constexpr (ABL & 4) { asm volatile("" :: "v"(x), "v"(y)); return; }
#pragma unroll
                    for (int q = 0; q < NPP; ++q) {
                        if constexpr (CPLX) {
                            T ar = arow[(q * CH + k) * 2], ai = arow[(q * CH + k) * 2 + 1];
                            accr[q][k] = tfma<T>(ar, x, accr[q][k]);
                            accr[q][k] = tfma<T>(-ai, y, accr[q][k]);
                            acci[q][k] = tfma<T>(ar, y, acci[q][k]);
                            acci[q][k] = tfma<T>(ai, x, acci[q][k]);
                        } else {
                            T a = arow[q * CH + k];
                            accr[q][k] = tfma<T>(a, x, accr[q][k]);
                            acci[q][k] = tfma<T>(a, y, acci[q][k]);
                        }
                    }
                });
            }
        }
        if (++since_flush == FLUSH_TILES) { flush(); since_flush = 0; }
    }
    if (since_flush > 0 || first) flush();
}

// ---------------------------------------------------------------------------------------
// backward (gradient w.r.t. psky)
// ---------------------------------------------------------------------------------------
template <typename T, int NPP, bool CPLX, int CH, int MODE, int PIX, int WPS = 1>
__global__ void __launch_bounds__(256, WPS)
fringe_bwd_kernel(FringeArgs A)
{
    using G = Geom<T, NPP, CPLX, CH>;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    double* b_lds = reinterpret_cast<double*>(smem_raw);                        // [3][TB]
    double* f_lds = b_lds + 3 * TB;                                             // [CH]
    T* g_lds = reinterpret_cast<T*>(f_lds + CH);                                // [TB][GSTRIDE]

    const int tid = threadIdx.x;
    const int bxi = blockIdx.x % A.nbx, tsi = blockIdx.x / A.nbx;
    const int k0 = blockIdx.y * CH;
    const int t = tsi / A.S;
    const int split = tsi % A.S;
    const int nk = min(CH, A.Nf - k0);
    const double nu_c = A.freq0_c + (double)(k0 + CH / 2) * A.dfreq_c;
    const double dnu = A.dfreq_c;

    if constexpr (MODE == MODE_DIRECT) {
        for (int i = tid; i < CH; i += blockDim.x)
            f_lds[i] = (i < nk) ? A.freqs[k0 + i] * (1.0 / 2.99792458e8) : 0.0;
    } else if constexpr (mode_is_nu(MODE)) {
        T* e_lds = reinterpret_cast<T*>(f_lds);
        for (int i = tid; i < CH; i += blockDim.x)
            e_lds[i] = (i < nk) ? (T)(6.283185307179586 * (A.freqs[k0 + i] * (1.0 / 2.99792458e8)
                                      - (A.freq0_c + (double)(k0 + i) * A.dfreq_c))) : T(0);
    }

    int p[PIX];
    double sx[PIX], sy[PIX], sz[PIX];
    const double* sd = A.sdir + (size_t)t * 3 * A.Pstride;
#pragma unroll
    for (int j = 0; j < PIX; ++j) {
        p[j] = (bxi * PIX + j) * blockDim.x + tid;
        const int pc = min(p[j], A.Pstride - 1);
        sx[j] = sd[pc]; sy[j] = sd[A.Pstride + pc]; sz[j] = sd[2 * (size_t)A.Pstride + pc];
    }

    T acc[PIX][NPP][CH * G::NC];
#pragma unroll
    for (int j = 0; j < PIX; ++j)
#pragma unroll
        for (int q = 0; q < NPP; ++q)
#pragma unroll
            for (int k = 0; k < CH * G::NC; ++k) acc[j][q][k] = T(0);

    const T* gvis = reinterpret_cast<const T*>(A.in);
    // S == 1: write gpsky in place (caller strides); S > 1: dense partial slab [split][t][q][f][p]
    const size_t plane = (size_t)NPP * A.Nf * A.Pstride * G::NC;
    const bool direct = (A.S == 1);
    T* dst = direct ? reinterpret_cast<T*>(A.out) + ((size_t)t * A.st_t + (size_t)A.mp * A.st_mp) * G::NC
                    : reinterpret_cast<T*>(A.ws) + ((size_t)split * A.Nt + t) * plane;
    const size_t d_pp = direct ? (size_t)A.st_pp : (size_t)A.Nf * A.Pstride;
    const size_t d_f = direct ? (size_t)A.st_f : (size_t)A.Pstride;
    bool first = true;
    auto flush = [&]() {
#pragma unroll
        for (int j = 0; j < PIX; ++j) {
            if (p[j] < A.Pstride) {
#pragma unroll
                for (int q = 0; q < NPP; ++q)
#pragma unroll
                    for (int k = 0; k < CH; ++k)
#pragma unroll
                        for (int c = 0; c < G::NC; ++c) {
                            if (k < nk) {
                                T* o = dst + ((size_t)q * d_pp + (size_t)(k0 + k) * d_f + p[j]) * G::NC + c;
                                T v = acc[j][q][k * G::NC + c];
                                if (!first) v += *o;
                                *o = v;
                            }
                            acc[j][q][k * G::NC + c] = T(0);
                        }
            }
        }
        first = false;
    };

    const int ntiles = (A.bl_cnt + TB - 1) / TB;
    const int tile_begin = split * A.tiles_per_split;
    const int tile_end = min(ntiles, tile_begin + A.tiles_per_split);
    int since_flush = 0;
    for (int tile = tile_begin; tile < tile_end; ++tile) {
        __syncthreads();
        const int s0 = tile * TB;
        for (int i = tid; i < 3 * TB; i += blockDim.x) {
            int bb = i % TB, d = i / TB;
            int slot = s0 + bb;
            double v = 0.0;
            if (slot < A.bl_cnt) {
                int b = A.bl_order ? A.bl_order[A.bl_off + slot] : A.bl_off + slot;
                v = A.sign * A.blvecs[3 * b + d];
            }
            b_lds[d * TB + bb] = v;
        }
        constexpr int NEL = TB * NPP * CH * 2;
        for (int i = tid; i < NEL; i += blockDim.x) {
            int e = i % (CH * 2);               // (k, c) within the chunk, k fastest in memory
            int q = (i / (CH * 2)) % NPP;
            int bb = i / (CH * 2 * NPP);
            int slot = s0 + bb;
            T v = T(0);
            if (slot < A.bl_cnt && (e >> 1) < nk) {
                int b = A.bl_order ? A.bl_order[A.bl_off + slot] : A.bl_off + slot;
                v = gvis[((((size_t)q * A.Nbl + b) * A.Nt + t) * A.Nf + k0) * 2 + e];
            }
            g_lds[bb * G::GSTRIDE + q * CH * 2 + e] = v;
        }
        __syncthreads();

        const int nb = min(TB, A.bl_cnt - s0);
        for (int bb = 0; bb < nb; ++bb) {
            const double bx = b_lds[bb], by = b_lds[TB + bb], bz = b_lds[2 * TB + bb];
            const T* grow = g_lds + bb * G::GSTRIDE;
#pragma unroll
            for (int j = 0; j < PIX; ++j) {
                const double tau = bx * sx[j] + by * sy[j] + bz * sz[j];
                fringe_chunk<T, CH, MODE>(tau, nu_c, dnu, f_lds, [&](int k, T x, T y) {
#pragma unroll
                    for (int q = 0; q < NPP; ++q) {
                        T gr = grow[(q * CH + k) * 2], gi = grow[(q * CH + k) * 2 + 1];
                        if constexpr (CPLX) {
                            // conj(F) * g = (x gr + y gi) + i (x gi - y gr)
                            acc[j][q][2 * k] = tfma<T>(gr, x, acc[j][q][2 * k]);
                            acc[j][q][2 * k] = tfma<T>(gi, y, acc[j][q][2 * k]);
                            acc[j][q][2 * k + 1] = tfma<T>(gi, x, acc[j][q][2 * k + 1]);
                            acc[j][q][2 * k + 1] = tfma<T>(-gr, y, acc[j][q][2 * k + 1]);
                        } else {
                            acc[j][q][k] = tfma<T>(gr, x, acc[j][q][k]);
                            acc[j][q][k] = tfma<T>(gi, y, acc[j][q][k]);
                        }
                    }
                });
            }
        }
        if (++since_flush == FLUSH_TILES) { flush(); since_flush = 0; }
    }
    if (since_flush > 0 || first) flush();
}

// materialised fringe (API parity with ArrayModel.gen_fringe; RIME itself never needs it):
// out[b, f, p] = exp(sign 2 pi i freq[f]/c  blvecs[b] . sdir[:, p]),  sdir is [3, sstride]
template <typename T>
__global__ void __launch_bounds__(256)
gen_fringe_kernel(const double* __restrict__ blvecs, const double* __restrict__ sdir,
                  const double* __restrict__ freqs, int Nbl, int Nf, int P, int sstride, double sign,
                  T* __restrict__ out)
{
    const int npb = (P + blockDim.x - 1) / blockDim.x;           // grid.x = Nbl * npb (grid.y is capped at 65535)
    const int p = (blockIdx.x % npb) * blockDim.x + threadIdx.x;
    const int b = blockIdx.x / npb;
    if (p >= P) return;
    const double tau = sign * (blvecs[3 * b] * sdir[p] + blvecs[3 * b + 1] * sdir[sstride + p]
                               + blvecs[3 * b + 2] * sdir[2 * (size_t)sstride + p]);
    for (int f = 0; f < Nf; ++f) {
        T s, c;
        sincos_turns(reduce_turns<T>(tau * (freqs[f] * (1.0 / 2.99792458e8))), s, c);
        T* o = out + (((size_t)b * Nf + f) * P + p) * 2;
        o[0] = c; o[1] = s;
    }
}

// deterministic reduction of S partial slabs: out[blk][i] = sum_s ws[s][blk][i]
template <typename T>
__global__ void reduce_partials_kernel(const T* __restrict__ ws, T* __restrict__ out,
                                       size_t len, int S, int nblk, size_t out_blk_stride,
                                       size_t out_offset)
{
    const size_t total = len * (size_t)nblk;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        size_t blk = i / len, e = i % len;
        T v = T(0);
        for (int s = 0; s < S; ++s) v += ws[((size_t)s * nblk + blk) * len + e];
        out[out_offset + blk * out_blk_stride + e] = v;
    }
}

// backward partial slabs [S][Nt][NPP][Nf][Pstride*NC] (dense) -> gpsky with caller strides
template <typename T>
__global__ void reduce_bwd_kernel(const T* __restrict__ ws, T* __restrict__ out, int S, int Nt, int NPP,
                                  int Nf, int PNC, int NC, long long st_t, long long st_pp,
                                  long long st_f, long long base)
{
    const size_t total = (size_t)Nt * NPP * Nf * PNC;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        size_t e = i % PNC, r = i / PNC;
        int f = r % Nf; r /= Nf;
        int q = r % NPP; int t = r / NPP;
        T v = T(0);
        for (int s = 0; s < S; ++s) v += ws[(size_t)s * total + i];
        out[(base + (size_t)t * st_t + (size_t)q * st_pp + (size_t)f * st_f) * NC + e] = v;
    }
}

// ---------------------------------------------------------------------------------------
// host-side dispatch
// ---------------------------------------------------------------------------------------
template <typename T> struct ChunkOf;   // channels per lane for each plane configuration
template <> struct ChunkOf<float>  { static constexpr int real1 = 32, real2 = 16, cplx1 = 16, real4 = 8, cplx4 = 8; };
template <> struct ChunkOf<double> { static constexpr int real1 = 16, real2 = 8, cplx1 = 8, real4 = 4, cplx4 = 4; };

static int pick_splits(long waves_unsplit, int max_splits)
{
    // 16 waves per SIMD over 1024 SIMDs = 4 rounds at the 4-waves/SIMD residency of the float
    // kernels: measured on MI355X (tools/fringe_lab.hip, C4 shape) 66.5 / 60.4 / 57.3 / 55.6 ms
    // for 2048 / 4096 / 8192 / 16384 waves -- extra rounds smooth the end-of-grid tail
    const long target = 16384;
    if (waves_unsplit >= target || max_splits <= 1) return 1;
    long s = (target + waves_unsplit - 1) / waves_unsplit;
    if (s > max_splits) s = max_splits;
    return (int)(s < 1 ? 1 : s);
}

struct Plan { int S; int tiles_per_split; int block; };

static Plan plan_fwd(int bl_cnt, int Nt, int Nf, int Pstride, int CH)
{
    Plan pl;
    pl.block = bl_cnt >= 256 ? 256 : ((bl_cnt + 63) / 64) * 64;
    const int nchunk = (Nf + CH - 1) / CH;
    const long nblk = (bl_cnt + pl.block - 1) / pl.block;
    const long waves = nblk * (pl.block / 64) * (long)nchunk * Nt;
    const int ntiles = Pstride / TP;
    pl.S = pick_splits(waves, std::max(1, ntiles / 4));      // keep >= 4 tiles (256 pixels) per split
    pl.tiles_per_split = (ntiles + pl.S - 1) / pl.S;
    pl.S = (ntiles + pl.tiles_per_split - 1) / pl.tiles_per_split;
    return pl;
}

static Plan plan_bwd(int bl_cnt, int Nt, int Nf, int Pstride, int CH, int PIX)
{
    Plan pl;
    const int per = 256 * PIX;
    pl.block = Pstride >= per ? 256 : (((Pstride + PIX - 1) / PIX + 63) / 64) * 64;
    const int nchunk = (Nf + CH - 1) / CH;
    const long nblk = (Pstride + pl.block * PIX - 1) / (pl.block * PIX);
    const long waves = nblk * (pl.block / 64) * (long)nchunk * Nt;
    // a model-pair group without baselines (bl_cnt == 0) still launches: its gradient plane must be written (zeros)
    const int ntiles = std::max(1, (bl_cnt + TB - 1) / TB);
    pl.S = pick_splits(waves, ntiles);
    pl.tiles_per_split = (ntiles + pl.S - 1) / pl.S;
    pl.S = (ntiles + pl.tiles_per_split - 1) / pl.tiles_per_split;
    return pl;
}

template <typename T, int NPP, bool CPLX, int CH>
static int launch_fwd_t(const FringeArgs& base, const int* mp_off, int mode, hipStream_t st,
                        size_t ws_bytes)
{
    using G = Geom<T, NPP, CPLX, CH>;
    // one split plan for the whole call (every model-pair group uses it), so a single
    // reduction over the full vis tensor finishes the job
    const Plan pl = plan_fwd(base.Nbl, base.Nt, base.Nf, base.Pstride, CH);
    const size_t vis_elems = (size_t)NPP * base.Nbl * base.Nt * base.Nf * 2;
    if (pl.S > 1 && ws_bytes < (size_t)pl.S * vis_elems * sizeof(T)) return RIME_EWORKSPACE;
    for (int g = 0; g < base.Nmp; ++g) {
        FringeArgs A = base;
        A.bl_off = mp_off[g];
        A.bl_cnt = mp_off[g + 1] - mp_off[g];
        A.mp = g;
        if (A.bl_cnt <= 0) continue;
        A.S = pl.S; A.tiles_per_split = pl.tiles_per_split;
        const int block = A.bl_cnt >= 256 ? 256 : ((A.bl_cnt + 63) / 64) * 64;
        A.nbx = (A.bl_cnt + block - 1) / block;
        dim3 grid((unsigned)A.nbx * A.Nt * A.S, (A.Nf + CH - 1) / CH, 1);
        size_t lds = (3 * TP + CH) * sizeof(double) + (size_t)TP * G::ASTRIDE * sizeof(T);
#define RIME_FWD(M) hipLaunchKernelGGL((fringe_fwd_kernel<T, NPP, CPLX, CH, M>), grid, dim3(block), lds, st, A)
        if constexpr (sizeof(T) == 4) {
            switch (mode) {
                case MODE_LIFT: RIME_FWD(MODE_LIFT); break;
                case MODE_LIFT_NU: RIME_FWD(MODE_LIFT_NU); break;
                case MODE_ROT_NU: RIME_FWD(MODE_ROT_NU); break;
                case MODE_DIRECT: RIME_FWD(MODE_DIRECT); break;
                default: RIME_FWD(MODE_ROT); break;
            }
        } else {
            switch (mode) {          // f64: the shear rotation is an f32 device (hardware rcp)
                case MODE_LIFT_NU: case MODE_ROT_NU: RIME_FWD(MODE_ROT_NU); break;
                case MODE_DIRECT: RIME_FWD(MODE_DIRECT); break;
                default: RIME_FWD(MODE_ROT); break;
            }
        }
#undef RIME_FWD
    }
    if (pl.S > 1) {
        int nb = (int)std::min<size_t>((vis_elems + 255) / 256, 4096);
        hipLaunchKernelGGL((reduce_partials_kernel<T>), dim3(nb), dim3(256), 0, st,
                           reinterpret_cast<const T*>(base.ws), reinterpret_cast<T*>(base.out),
                           vis_elems, pl.S, 1, (size_t)0, (size_t)0);
    }
    return check_launch();
}

template <typename T, int NPP, bool CPLX, int CH>
static int launch_bwd_t(const FringeArgs& base, const int* mp_off, int mode, hipStream_t st,
                        size_t ws_bytes)
{
    using G = Geom<T, NPP, CPLX, CH>;
    constexpr int PIX = 1;
    for (int g = 0; g < base.Nmp; ++g) {
        FringeArgs A = base;
        A.bl_off = mp_off[g];
        A.bl_cnt = mp_off[g + 1] - mp_off[g];
        A.mp = g;
        Plan pl = plan_bwd(A.bl_cnt, A.Nt, A.Nf, A.Pstride, CH, PIX);
        A.S = pl.S < 1 ? 1 : pl.S; A.tiles_per_split = pl.tiles_per_split;
        const size_t plane = (size_t)NPP * A.Nf * A.Pstride * G::NC;
        if (A.S > 1 && ws_bytes < (size_t)A.S * A.Nt * plane * sizeof(T)) return RIME_EWORKSPACE;
        A.nbx = (A.Pstride + pl.block * PIX - 1) / (pl.block * PIX);
        dim3 grid((unsigned)A.nbx * A.Nt * A.S, (A.Nf + CH - 1) / CH, 1);
        size_t lds = (3 * TB + CH) * sizeof(double) + (size_t)TB * G::GSTRIDE * sizeof(T);
        // 4 waves/SIMD (<=128 VGPRs, a few spilled dwords) beat 3 at 136 VGPRs: 66.4 vs 71.9 ms (lab)
        constexpr int WPS = (sizeof(T) == 4 && NPP * (CPLX ? 2 : 1) * CH <= 32) ? 4 : 1;
#define RIME_BWD(M, W) hipLaunchKernelGGL((fringe_bwd_kernel<T, NPP, CPLX, CH, M, PIX, W>), grid, dim3(pl.block), lds, st, A)
        if constexpr (sizeof(T) == 4) {
            switch (mode) {
                case MODE_LIFT: RIME_BWD(MODE_LIFT, WPS); break;
                case MODE_LIFT_NU: RIME_BWD(MODE_LIFT_NU, WPS); break;
                case MODE_ROT_NU: RIME_BWD(MODE_ROT_NU, WPS); break;
                case MODE_DIRECT: RIME_BWD(MODE_DIRECT, 1); break;
                default: RIME_BWD(MODE_ROT, WPS); break;
            }
        } else {
            switch (mode) {
                case MODE_LIFT_NU: case MODE_ROT_NU: RIME_BWD(MODE_ROT_NU, 1); break;
                case MODE_DIRECT: RIME_BWD(MODE_DIRECT, 1); break;
                default: RIME_BWD(MODE_ROT, 1); break;
            }
        }
#undef RIME_BWD
        if (A.S > 1) {
            int nb = (int)std::min<size_t>((plane * A.Nt + 255) / 256, 4096);
            hipLaunchKernelGGL((reduce_bwd_kernel<T>), dim3(nb), dim3(256), 0, st,
                               reinterpret_cast<const T*>(A.ws), reinterpret_cast<T*>(A.out),
                               A.S, A.Nt, NPP, A.Nf, A.Pstride * G::NC, G::NC,
                               A.st_t, A.st_pp, A.st_f, (long long)g * A.st_mp);
        }
    }
    return check_launch();
}

template <typename T>
static int dispatch(bool backward, const FringeArgs& A, const int* mp_off, int Npp, int cplx,
                    int mode, hipStream_t st, size_t ws_bytes)
{
    using C = ChunkOf<T>;
#define RIME_GO(NPP, CP, CH)                                                              \
    return backward ? launch_bwd_t<T, NPP, CP, CH>(A, mp_off, mode, st, ws_bytes)          \
                    : launch_fwd_t<T, NPP, CP, CH>(A, mp_off, mode, st, ws_bytes)
    if (Npp == 1 && !cplx) { RIME_GO(1, false, C::real1); }
    if (Npp == 2 && !cplx) { RIME_GO(2, false, C::real2); }
    if (Npp == 1 && cplx)  { RIME_GO(1, true, C::cplx1); }
    if (Npp == 4 && !cplx) { RIME_GO(4, false, C::real4); }
    if (Npp == 4 && cplx)  { RIME_GO(4, true, C::cplx4); }
#undef RIME_GO
    return RIME_EUNSUPPORTED;
}

static int chunk_for(int dtype, int Npp, int cplx)
{
    if (dtype == RIME_F32) {
        using C = ChunkOf<float>;
        return Npp == 1 ? (cplx ? C::cplx1 : C::real1) : Npp == 2 ? C::real2 : (cplx ? C::cplx4 : C::real4);
    }
    using C = ChunkOf<double>;
    return Npp == 1 ? (cplx ? C::cplx1 : C::real1) : Npp == 2 ? C::real2 : (cplx ? C::cplx4 : C::real4);
}

} // namespace rime

using namespace rime;

extern "C" size_t rime_fringe_sum_workspace(int dtype, int Nbl, int Nt, int Nf, int Pstride,
                                            int Nmp, int Npp, int psky_complex, int backward)
{
    const size_t tsz = dtype == RIME_F64 ? 8 : 4;
    const int CH = chunk_for(dtype, Npp, psky_complex);
    (void)Nmp;
    if (!backward) {
        Plan pl = plan_fwd(Nbl, Nt, Nf, Pstride, CH);
        return pl.S <= 1 ? 0 : (size_t)pl.S * Npp * Nbl * Nt * Nf * 2 * tsz;
    }
    // every beam-model-pair group plans its own baseline splits (plan_bwd(bl_cnt)): the split count is not
    // monotone in the group size (it is rounded to whole tiles per split), but never exceeds the un-rounded
    // pick for the largest possible group -- that bound sizes the workspace
    const int PIX = 1;
    const int block = Pstride >= 256 * PIX ? 256 : (((Pstride + PIX - 1) / PIX + 63) / 64) * 64;
    const long nblk = (Pstride + block * PIX - 1) / (block * PIX);
    const long waves = nblk * (block / 64) * (long)((Nf + CH - 1) / CH) * Nt;
    const int S = pick_splits(waves, (Nbl + TB - 1) / TB);
    return S <= 1 ? 0 : (size_t)S * Nt * Npp * Nf * Pstride * (psky_complex ? 2 : 1) * tsz;
}

static int fringe_common(bool backward, int dtype, const double* blvecs, const double* sdir,
                         const double* freqs, const void* in, const int* mp_off, const int* bl_order,
                         int Nbl, int Nt, int Nf, int Pstride, int Nmp, int Npp, int cplx, int sign,
                         int uniform, double freq0, double dfreq, double max_blen,
                         const long long* strides, void* out, void* ws, size_t ws_bytes, void* stream)
{
    if (!blvecs || !sdir || !freqs || !in || !out || !mp_off) return RIME_EINVAL;
    if (Nbl <= 0 || Nt <= 0 || Nf <= 0 || Pstride <= 0 || Nmp <= 0) return RIME_EINVAL;
    if (Pstride % TP != 0) return RIME_EINVAL;
    if (!(Npp == 1 || Npp == 2 || Npp == 4)) return RIME_EINVAL;
    if (Npp == 2 && cplx) return RIME_EUNSUPPORTED;
    if (sign != 1 && sign != -1) return RIME_EINVAL;
    if (mp_off[0] != 0 || mp_off[Nmp] != Nbl) return RIME_EINVAL;
    if (dtype != RIME_F32 && dtype != RIME_F64) return RIME_EINVAL;
    FringeArgs A{};
    A.blvecs = blvecs; A.sdir = sdir; A.freqs = freqs; A.in = in; A.out = out; A.ws = ws;
    A.bl_order = bl_order;
    A.Nbl = Nbl; A.Nt = Nt; A.Nf = Nf; A.Pstride = Pstride; A.Nmp = Nmp;
    if (strides) {
        A.st_t = strides[0]; A.st_mp = strides[1]; A.st_pp = strides[2]; A.st_f = strides[3];
        if ((Nf > 1 && A.st_f < Pstride) || (Nt > 1 && A.st_t <= 0) || (Npp > 1 && A.st_pp <= 0) ||
            (Nmp > 1 && A.st_mp <= 0)) return RIME_EINVAL;
    } else {
        A.st_f = Pstride; A.st_pp = (long long)Nf * Pstride; A.st_mp = A.st_pp * Npp; A.st_t = A.st_mp * Nmp;
    }
    A.sign = (double)sign;
    A.freq0_c = freq0 / 2.99792458e8;
    A.dfreq_c = dfreq / 2.99792458e8;
    int mode = MODE_DIRECT;
    if (uniform == 1 || uniform == 2) {
        const bool nu = (uniform == 2);
        mode = nu ? MODE_ROT_NU : MODE_ROT;
        // lifting needs |step| comfortably below half a turn: tan(pi*step) stays O(1)
        if (max_blen > 0 && max_blen * fabs(A.dfreq_c) < 0.3) mode = nu ? MODE_LIFT_NU : MODE_LIFT;
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == RIME_F32) return dispatch<float>(backward, A, mp_off, Npp, cplx, mode, st, ws_bytes);
    return dispatch<double>(backward, A, mp_off, Npp, cplx, mode, st, ws_bytes);
}

extern "C" int rime_fringe_sum_fwd(int dtype, const double* blvecs, const double* sdir,
                                   const double* freqs, const void* psky,
                                   const int* mp_offsets_host, const int* bl_order,
                                   int Nbl, int Nt, int Nf, int Pstride, int Nmp, int Npp,
                                   int psky_complex, int sign, int freq_uniform_host,
                                   double freq0_host, double dfreq_host, double max_blen_host,
                                   const long long* psky_strides_host,
                                   void* vis, void* workspace, size_t workspace_bytes, void* stream)
{
    return fringe_common(false, dtype, blvecs, sdir, freqs, psky, mp_offsets_host, bl_order, Nbl, Nt,
                         Nf, Pstride, Nmp, Npp, psky_complex, sign, freq_uniform_host, freq0_host,
                         dfreq_host, max_blen_host, psky_strides_host, vis, workspace, workspace_bytes, stream);
}

extern "C" int rime_fringe_sum_bwd(int dtype, const double* blvecs, const double* sdir,
                                   const double* freqs, const void* gvis,
                                   const int* mp_offsets_host, const int* bl_order,
                                   int Nbl, int Nt, int Nf, int Pstride, int Nmp, int Npp,
                                   int psky_complex, int sign, int freq_uniform_host,
                                   double freq0_host, double dfreq_host, double max_blen_host,
                                   const long long* gpsky_strides_host,
                                   void* gpsky, void* workspace, size_t workspace_bytes, void* stream)
{
    return fringe_common(true, dtype, blvecs, sdir, freqs, gvis, mp_offsets_host, bl_order, Nbl, Nt,
                         Nf, Pstride, Nmp, Npp, psky_complex, sign, freq_uniform_host, freq0_host,
                         dfreq_host, max_blen_host, gpsky_strides_host, gpsky, workspace, workspace_bytes, stream);
}

extern "C" int rime_gen_fringe(int dtype, const double* blvecs, const double* sdir,
                               const double* freqs, int Nbl, int Nf, int P, int sdir_stride,
                               int sign, void* out, void* stream)
{
    if (!blvecs || !sdir || !freqs || !out) return RIME_EINVAL;
    if (Nbl <= 0 || Nf <= 0 || P <= 0 || sdir_stride < P || (sign != 1 && sign != -1)) return RIME_EINVAL;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    dim3 grid((unsigned)((P + 255) / 256) * Nbl), block(256);
    if (dtype == RIME_F32)
        hipLaunchKernelGGL((gen_fringe_kernel<float>), grid, block, 0, st, blvecs, sdir, freqs, Nbl, Nf, P,
                           sdir_stride, (double)sign, reinterpret_cast<float*>(out));
    else if (dtype == RIME_F64)
        hipLaunchKernelGGL((gen_fringe_kernel<double>), grid, block, 0, st, blvecs, sdir, freqs, Nbl, Nf, P,
                           sdir_stride, (double)sign, reinterpret_cast<double*>(out));
    else return RIME_EINVAL;
    return check_launch();
}
