// cal.hip -- per-visibility gain application, the post-RIME epilogue of SURVEY section 8(f) item 3:
//     V'[., ., b, t, f] = G1 V G2^dagger,   G1 = gains[., ., a1(b), t, f], G2 = gains[., ., a2(b), t, f]
// (calibration._apply_cal, calibration.py:2412-2487, complex visibilities, no undo / covariance):
//   NP = 1                : g1 conj(g2) v
//   NP = 2, diagonal mode : the two diagonal products only, off-diagonals of the result zero
//                           (linalg.diag_matmul, linalg.py:116-149)
//   NP = 2, full          : 2x2 products g1 v g2^dagger
// One pass over the visibility tensor instead of two gain gathers (each the size of vis), a conj and
// two products.  Backward in one pass too: gvis = G1^dagger gout G2 and the per-baseline gain
// gradients  d1 = gout G2 V^dagger  (for antenna a1),  d2 = gout^dagger G1 V  (for a2); their
// reduction over baselines (and over broadcast time / channel axes) is a dense 0/1-matrix product on
// the host side (deterministic).  Gains broadcast over time / channel through element strides (0).
#include <hip/hip_runtime.h>
#include "rime_common.h"

namespace rime {

template <typename T> struct cx { T re, im; };
template <typename T> __device__ __forceinline__ cx<T> cmul(cx<T> a, cx<T> b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
template <typename T> __device__ __forceinline__ cx<T> cmulc(cx<T> a, cx<T> b) { return {a.re * b.re + a.im * b.im, a.im * b.re - a.re * b.im}; }   // a conj(b)
template <typename T> __device__ __forceinline__ cx<T> cconj(cx<T> a) { return {a.re, -a.im}; }
template <typename T> __device__ __forceinline__ cx<T> cadd(cx<T> a, cx<T> b) { return {a.re + b.re, a.im + b.im}; }

struct CalArgs {
    const void* vis; const void* gains; const void* gout;      // vis / gout [NP,NP,Nbl,Nt,Nf,2]; gains [NP,NP,Nant,Ntg,Nfg,2]
    const int* a1; const int* a2;                              // [Nbl]
    void* out; void* gvis; void* d1; void* d2;                 // all [NP,NP,Nbl,Nt,Nf,2]
    int Nbl, Nt, Nf, Nant;
    long long gst_a, gst_t, gst_f, gst_p;                      // gain strides (complex elements): antenna, time, channel, pol entry
};

template <typename T, int NP, bool DIAG, bool BWD>
__global__ void __launch_bounds__(256)
apply_cal_kernel(CalArgs A)
{
    const size_t n = (size_t)A.Nbl * A.Nt * A.Nf;
    const size_t plane = n;                                    // complex elements per pol entry of vis
    const cx<T>* vis = reinterpret_cast<const cx<T>*>(A.vis);
    const cx<T>* gains = reinterpret_cast<const cx<T>*>(A.gains);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int f = (int)(i % A.Nf);
        const int t = (int)((i / A.Nf) % A.Nt);
        const int b = (int)(i / ((size_t)A.Nf * A.Nt));
        const size_t o1 = (size_t)A.a1[b] * A.gst_a + (size_t)t * A.gst_t + (size_t)f * A.gst_f;
        const size_t o2 = (size_t)A.a2[b] * A.gst_a + (size_t)t * A.gst_t + (size_t)f * A.gst_f;
        cx<T> g1[NP][NP], g2[NP][NP], v[NP][NP];
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                g1[p][q] = gains[o1 + (size_t)(p * NP + q) * A.gst_p];
                g2[p][q] = gains[o2 + (size_t)(p * NP + q) * A.gst_p];
                v[p][q] = vis[(size_t)(p * NP + q) * plane + i];
            }
        if constexpr (!BWD) {
            cx<T>* out = reinterpret_cast<cx<T>*>(A.out);
            if constexpr (NP == 1 || DIAG) {
#pragma unroll
                for (int p = 0; p < NP; ++p)
#pragma unroll
                    for (int q = 0; q < NP; ++q)
                        out[(size_t)(p * NP + q) * plane + i] = (p == q) ? cmul(cmulc(g1[p][p], g2[p][p]), v[p][p]) : cx<T>{T(0), T(0)};
            } else {
#pragma unroll
                for (int a = 0; a < NP; ++a)
#pragma unroll
                    for (int d = 0; d < NP; ++d) {
                        cx<T> acc = {T(0), T(0)};
#pragma unroll
                        for (int bb = 0; bb < NP; ++bb)
#pragma unroll
                            for (int c = 0; c < NP; ++c) acc = cadd(acc, cmulc(cmul(g1[a][bb], v[bb][c]), g2[d][c]));
                        out[(size_t)(a * NP + d) * plane + i] = acc;
                    }
            }
        } else {
            const cx<T>* gout = reinterpret_cast<const cx<T>*>(A.gout);
            cx<T>* gvis = reinterpret_cast<cx<T>*>(A.gvis);
            cx<T>* d1 = reinterpret_cast<cx<T>*>(A.d1);
            cx<T>* d2 = reinterpret_cast<cx<T>*>(A.d2);
            cx<T> go[NP][NP];
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int q = 0; q < NP; ++q) go[p][q] = gout[(size_t)(p * NP + q) * plane + i];
            if constexpr (NP == 1 || DIAG) {
#pragma unroll
                for (int p = 0; p < NP; ++p)
#pragma unroll
                    for (int q = 0; q < NP; ++q) {
                        const size_t k = (size_t)(p * NP + q) * plane + i;
                        if (p == q) {
                            // out = g1 conj(g2) v:  gvis = conj(g1 conj(g2)) go;  d1 = go conj(conj(g2) v);  d2 = conj(go) g1 v
                            gvis[k] = cmul(cconj(cmulc(g1[p][p], g2[p][p])), go[p][p]);
                            d1[k] = cmulc(go[p][p], cmulc(v[p][p], g2[p][p]));
                            d2[k] = cmul(cconj(go[p][p]), cmul(g1[p][p], v[p][p]));
                        } else {
                            gvis[k] = {T(0), T(0)}; d1[k] = {T(0), T(0)}; d2[k] = {T(0), T(0)};
                        }
                    }
            } else {
                // gvis = G1^H go G2;  d1 = go (V G2^H)^H = go G2 V^H;  d2 = (G1 V)^H go -> grad of G2 is d2^H... see ops
                cx<T> W[NP][NP], X[NP][NP];          // W = V G2^H (b, d);  X = G1 V (a, c)
#pragma unroll
                for (int bb = 0; bb < NP; ++bb)
#pragma unroll
                    for (int d = 0; d < NP; ++d) {
                        cx<T> acc = {T(0), T(0)};
#pragma unroll
                        for (int c = 0; c < NP; ++c) acc = cadd(acc, cmulc(v[bb][c], g2[d][c]));
                        W[bb][d] = acc;
                    }
#pragma unroll
                for (int a = 0; a < NP; ++a)
#pragma unroll
                    for (int c = 0; c < NP; ++c) {
                        cx<T> acc = {T(0), T(0)};
#pragma unroll
                        for (int bb = 0; bb < NP; ++bb) acc = cadd(acc, cmul(g1[a][bb], v[bb][c]));
                        X[a][c] = acc;
                    }
#pragma unroll
                for (int p = 0; p < NP; ++p)
#pragma unroll
                    for (int q = 0; q < NP; ++q) {
                        const size_t k = (size_t)(p * NP + q) * plane + i;
                        cx<T> gv = {T(0), T(0)}, e1 = {T(0), T(0)}, e2 = {T(0), T(0)};
#pragma unroll
                        for (int a = 0; a < NP; ++a)
#pragma unroll
                            for (int d = 0; d < NP; ++d)
                                gv = cadd(gv, cmul(cmul(cconj(g1[a][p]), go[a][d]), g2[d][q]));     // (G1^H go G2)[p][q]
#pragma unroll
                        for (int d = 0; d < NP; ++d) e1 = cadd(e1, cmulc(go[p][d], W[q][d]));        // (go W^H)[p][q]: grad G1
#pragma unroll
                        for (int a = 0; a < NP; ++a) e2 = cadd(e2, cmul(cconj(go[a][p]), X[a][q]));  // (go^H X)[p][q]: grad G2
                        gvis[k] = gv; d1[k] = e1; d2[k] = e2;
                    }
            }
        }
    }
}

template <typename T, bool BWD>
static int cal_launch(const CalArgs& A, int NP, int diag, hipStream_t st)
{
    const size_t n = (size_t)A.Nbl * A.Nt * A.Nf;
    const int nb = (int)std::min<size_t>((n + 255) / 256, 16384);
    if (NP == 1) hipLaunchKernelGGL((apply_cal_kernel<T, 1, true, BWD>), dim3(nb), dim3(256), 0, st, A);
    else if (diag) hipLaunchKernelGGL((apply_cal_kernel<T, 2, true, BWD>), dim3(nb), dim3(256), 0, st, A);
    else hipLaunchKernelGGL((apply_cal_kernel<T, 2, false, BWD>), dim3(nb), dim3(256), 0, st, A);
    return check_launch();
}

} // namespace rime

using namespace rime;

static bool cal_args_ok(int NP, int Nbl, int Nt, int Nf, int Nant)
{
    return (NP == 1 || NP == 2) && Nbl > 0 && Nt > 0 && Nf > 0 && Nant > 0;
}

extern "C" int rime_apply_cal_fwd(int dtype, int NP, int diag, const void* vis, const void* gains, const int* a1,
                                  const int* a2, int Nbl, int Nt, int Nf, int Nant, long long gst_p, long long gst_a,
                                  long long gst_t, long long gst_f, void* out, void* stream)
{
    if (!vis || !gains || !a1 || !a2 || !out || !cal_args_ok(NP, Nbl, Nt, Nf, Nant)) return RIME_EINVAL;
    CalArgs A{};
    A.vis = vis; A.gains = gains; A.a1 = a1; A.a2 = a2; A.out = out;
    A.Nbl = Nbl; A.Nt = Nt; A.Nf = Nf; A.Nant = Nant;
    A.gst_p = gst_p; A.gst_a = gst_a; A.gst_t = gst_t; A.gst_f = gst_f;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == RIME_F32) return cal_launch<float, false>(A, NP, diag, st);
    if (dtype == RIME_F64) return cal_launch<double, false>(A, NP, diag, st);
    return RIME_EINVAL;
}

extern "C" int rime_apply_cal_bwd(int dtype, int NP, int diag, const void* vis, const void* gains, const void* gout,
                                  const int* a1, const int* a2, int Nbl, int Nt, int Nf, int Nant, long long gst_p,
                                  long long gst_a, long long gst_t, long long gst_f, void* gvis, void* d1, void* d2,
                                  void* stream)
{
    if (!vis || !gains || !gout || !a1 || !a2 || !gvis || !d1 || !d2 || !cal_args_ok(NP, Nbl, Nt, Nf, Nant)) return RIME_EINVAL;
    CalArgs A{};
    A.vis = vis; A.gains = gains; A.gout = gout; A.a1 = a1; A.a2 = a2; A.gvis = gvis; A.d1 = d1; A.d2 = d2;
    A.Nbl = Nbl; A.Nt = Nt; A.Nf = Nf; A.Nant = Nant;
    A.gst_p = gst_p; A.gst_a = gst_a; A.gst_t = gst_t; A.gst_f = gst_f;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == RIME_F32) return cal_launch<float, true>(A, NP, diag, st);
    if (dtype == RIME_F64) return cal_launch<double, true>(A, NP, diag, st);
    return RIME_EINVAL;
}
