// fringe_mfma.hip -- antenna-factored fringe sum on the matrix cores (gfx950), forward and backward.
//
// For baselines that are antenna pairs, b_ij = r_j - r_i, the fringe factorises:
//     exp(2 pi i nu b_ij.s / c) = E_j conj(E_i),   E_a[f,p] = exp(2 pi i nu_f r_a.s_p / c)
// so for every (time, channel) the visibilities of ALL pairs are one Hermitian rank-P update
//     V[i,j] = sum_p conj(E_i[p]) * (A[p] E_j[p]),        A = one real plane of psky[t,f,:]
// i.e. a complex GEMM with M = N = Nant, K = P, batched over (t, f) -- the "dense
// (Nvis x Npix) . Npix contraction" of the north star, at 1/Nant of the exponentials of the
// baseline formulation.  (Reference arithmetic replaced: the same lines as fringe.hip,
// telescope_model.py:310-358 + rime_model.py:423-429.)
//
// Precision: f32 operands are split into two f16 halves (hi + lo, 21 significant bits); the three
// cross products hi*hi + hi*lo + lo*hi run on v_mfma_f32_32x32x16_f16 with f32 accumulation
// (the dropped lo*lo term is 2^-22 relative).  psky rows are pre-scaled by a power of two per
// (t, f) so that the f16 range is used (`scale` input).
//
// Work decomposition: block = one (t, f, pixel split); 4 waves; the upper-triangular 32x32 tiles
// of the Nant x Nant (<= 128 x 128) output are dealt to the waves; per panel of 32 pixels the
// block generates the operand images once into LDS and every wave runs its MFMAs from LDS
// fragments (details at the kernels).  Larger arrays: groups of 128 antennas, diagonal blocks (this
// kernel) + cross blocks (8 waves, 16 tiles); multi-pol / complex psky: one launch per real plane.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <type_traits>
#include "rime_common.h"

namespace rime {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr int MF_NA = 128;                   // antennas per block (4 x 4 tiles)
constexpr int MF_SPLIT_PIX = 16384;           // pixels per block (bounds the f32 MFMA accumulation chain)

struct AntArgs {
    const double* antpos;      // [Nant, 3]
    const double* sdir;        // [Nt, 3, Pstride]
    const double* freqs;       // [Nf]
    const float* psky;         // strided [t][f][p]
    const float* scale;        // [Nt, Nf] power-of-two pre-scale of psky rows
    const float* rowmin;       // [Nt, Nf] min of each psky row, or NULL: rows with min >= 0 skip the sign masks
    const int* pair_direct;    // [128*128] baseline slot receiving V[i,j], or -1
    const int* pair_conj;      // [128*128] baseline slot receiving conj(V[i,j]), or -1
    float* vis;                // [Nbl, Nt, Nf, 2]
    float* ws;                 // partial slabs when S > 1
    int Nant, Nbl, Nt, Nf, Pstride;
    int S, panels_per_split;
    long long st_t, st_f, st_p;   // element strides of psky: time, channel, pixel (2 = one plane of a complex buffer)
    double sign;
    float imsign;              // complex single-pass blocks: +1, or -1 to contract with conj(psky) (swapped groups)
    int mirror;                // diagonal blocks, real psky: bit g set = rows 16 g + 8 .. 16 g + 15 hold the MIRROR antennas of rows
                               // 16 g .. 16 g + 7 (r' - c = -(r - c) about the centre the positions are measured from): their
                               // phasors are the complex conjugates and are not evaluated again (round 5, see MIRROR PAIRS below)
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

// the same split for values that are not products: the residual a - hi is one mixed-precision FMA
// (a * 1.0 - hi, the f16 half read in place) instead of v_cvt_f32_f16 + v_sub_f32 -- the compiler
// only forms v_fma_mix when there is a multiply to fuse
__device__ __forceinline__ void split2_plain(float a, float b, uint32_t& hi, uint32_t& lo)
{
    auto h = __builtin_amdgcn_cvt_pkrtz(a, b);
    hi = __builtin_bit_cast(uint32_t, h);
    float ra, rb;
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(ra) : "v"(a), "v"(hi));
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(rb) : "v"(b), "v"(hi));
    lo = pack_rtz(ra, rb);
}

// Keeps a float a scalar computation of its own: the compiler's SLP pass pairs independent f32 adds / multiplies / FMAs into
// v_pk_*_f32.  Round 5: in the conjugate-pair backward -- the first kernel here whose blocks SHARE a CU, so that one block stages its
// G planes while another block's waves stream MFMAs on the same SIMDs -- the build whose staging arithmetic the compiler had
// vectorised (65 v_pk_add_f32: sums of two freshly loaded gradients) produced wrong planes in a fraction of the blocks that
// were dispatched late, different from run to run; alone on the CU, or with scalar adds, the same code is exact
// (tools/debug_pair_bwd.py, profiles/r05/pair_form.txt 3 and 8).  A hand-written v_pk_add_f32 in the same place, and the
// compiler's add / subtract / select sequence verbatim in inline asm, are exact too: the mechanism is open.  With round 2's
// stale packed reads of matrix-core results (rime_common.h) it is the second sighting of compiler-packed f32 arithmetic going
// wrong beside a busy matrix pipe, so the pair kernels contain NO packed f32 instruction (the build scans for them): a fence
// around an observed failure, not an explanation of it.
__device__ __forceinline__ void keep_scalar(float& x) { asm("" : "+v"(x)); }

// the split of two PRODUCTS (a0 b0, a1 b1) that are also read as f32 (p0, p1): the residuals a b - hi as one mixed-precision FMA
// each on the exact product -- what the compiler makes of split2(a0 * b0, a1 * b1, ..) when the products have no other reader
// (with one, it subtracts the rounded product: v_cvt_f32_f16 + v_sub_f32, two instructions per value)
__device__ __forceinline__ void split2_prod(float a0, float b0, float a1, float b1, float& p0, float& p1, uint32_t& hi, uint32_t& lo)
{
    p0 = a0 * b0;
    keep_scalar(p0);
    p1 = a1 * b1;
    hi = pack_rtz(p0, p1);
    float ra, rb;
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(ra) : "v"(a0), "v"(b0), "v"(hi));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(rb) : "v"(a1), "v"(b1), "v"(hi));
    lo = pack_rtz(ra, rb);
}

// Phase of a phasor in turns: a.s = ax sx + ay sy + az sz in float64 (antenna coordinates pre-multiplied by sign nu / c), reduced to
// its fraction as a float32 for v_sin_f32 / v_cos_f32 (a fixed-point reduction in the low mantissa bits saves two instructions
// per phasor for 2.2 x the phase noise: measured, not adopted -- profiles/r04/phase_magic_ab.txt, tools/lab/).
__device__ __forceinline__ double phase3(double ax, double sx, double ay, double sy, double az, double sz)
{
    return ax * sx + ay * sy + az * sz;
}
__device__ __forceinline__ float turn_frac(double ph) { return (float)__builtin_amdgcn_fract(ph); }

// FLAT (round 5, the conjugate-pair kernels): every row's z coordinate is zero (a coplanar array measured from a centre in its
// plane -- the layouts simulations run on), so the third term of the phase is exactly zero and is not evaluated: one f64 FMA
// less per phasor (3 % of a headline step; a licence stated by the caller, like `mirror`)
template <bool FLAT>
__device__ __forceinline__ double phase_of(double ax, double sx, double ay, double sy, double az, double sz)
{
    if constexpr (FLAT) return ax * sx + ay * sy;
    else return phase3(ax, sx, ay, sy, az, sz);
}

__device__ __forceinline__ f16x8 as_frag(const uint4& v) { return __builtin_bit_cast(f16x8, v); }

// ---------------------------------------------------------------------------------------
// forward kernel
//
// Measured on gfx950 (tools/overlap_lab.hip, profiles/r01/overlap_lab.txt): VALU instructions and
// MFMAs issued on one SIMD do NOT overlap -- a wave streaming v_mfma_f32_32x32x16_f16 starves the
// VALU work of every other wave on its SIMD (s_setprio, s_nop padding and de-phasing co-resident
// blocks change nothing), and inside one wave only ~3 VALU ops hide under each MFMA.  Kernel time
// is therefore (MFMA issue time) + (VALU issue time), and the design minimises both:
//   * symmetric weighting: L = B = sqrt(|psky| scale) E, so ONE operand image pair (f16 hi, lo) is
//     generated per (antenna, pixel) instead of two; the sign of psky is applied as an XOR mask on
//     the row-tile fragments (per wave: 16 v_xor per distinct row tile and panel);
//   * re and im live in separate K-planes ([antenna][32 px re | 32 px im]), K = 16 pixels per
//     MFMA.  Vr = Lr.Br + Li.Bi,  Vi = Lr.Bi - Li.Br with the two Vi products kept in separate
//     accumulators and subtracted in the epilogue: no operand rotation / negation work at all;
//   * the 10 upper-triangular 32x32 tiles are dealt to the 4 waves as 20 (tile, re|im) units,
//     5 each (a 3/3/2/2 deal of whole tiles idles 17 % of the pipe), in a tile order that lets two
//     of the four waves touch a single row tile; every wave gets both units of one diagonal tile;
//   * 33..64 antennas: a wave generates one 16-pixel half of the panel for every second row octet
//     (half the pointing-vector loads and registers) and skips octets that are all padding (6-7 %);
//   * diagonal tiles multiply an image by itself, so their lo x hi products are the transposes of
//     the hi x lo ones and Li.Br is the transpose of Lr.Bi: 7 MFMAs per K step instead of 12 (25
//     per wave and K step instead of 30), the transposes taken once in the epilogue through LDS
//     (10.7 -> 9.7 ms; re-using the diagonal tile's B fragments as L fragments instead of a
//     second LDS read is 2 % slower);
//   * a thread generates 2 adjacent pixels x TA antennas whose coordinates (pre-multiplied by
//     sign nu/c) stay in registers: per pair 3 f64 FMA + fract + cvt + sin + cos + mul and half
//     a hi/lo split; packed (p, p+1) f16 pairs go out as conflict-free ds_write_b32;
//   * LDS images are double buffered, one barrier per 32-pixel panel (two K steps): 16-pixel
//     panels are 7 % slower (barrier + first-fragment latency per MFMA burst);
//   * tried and dropped: v_fma_mixlo/hi_f16 for the split (fewer instructions, but a serial
//     dependency through the half-register writes: 2 % slower); v_pk_mul_f32 for the weight
//     products (packed f32 costs two issue slots: 2 % slower); the f64 "magic number" range
//     reduction (1 % faster, 4 % more phase noise); a per-K-step unit deal for <= 64 antennas (no
//     gain: those shapes are bound by operand generation); > 8192 pixels per split (slower tail,
//     3x the f32 accumulation error); a uniform branch that skips the sign masks (spills); one
//     accumulator per imaginary-part unit with a negated Li fragment (4 % slower, the freed
//     registers only change the schedule); antenna coordinates in LDS instead of registers (2 % slower);
//     one LDS buffer with two barriers per 32- or 64-pixel panel and a rolling pixel prefetch (1-2 %
//     slower); same-wave interleave of the next panel's generation with the MFMAs (the ~3 VALU slots
//     that hide under each MFMA of the same wave, forced with sched_group_barrier on a branch-free
//     generate): 40 spilled registers at 2 waves per SIMD, and with 1 wave per SIMD (accumulators in
//     AGPRs, no spills) LDS latency and barriers are exposed: 13.0-13.5 ms against 10.7.
// Every block covers at most MF_SPLIT_PIX pixels and STORES its result (no read-modify-write):
// f32 accumulation inside the MFMA chain stays below eps*sqrt(1024) (round 4: 16 384 pixels per block instead of 8192 --
// half the slab traffic, C4 forward - 1 %, measured error against float64 2.3e-6 -> 2.8e-6 of the maximum, identical for
// 24 576: profiles/r04/split_size.txt), and the pixel splits are
// summed (and transposed into the result layout) by reduce_vis_kernel in a fixed order.
// History (C4 shape, 128 antennas, 256 channels x 2 times, 98304 px): interleaved (re,im) K layout
// with separate L/B images, rot90 on the fly and whole-tile deal: 13.4 ms; this kernel: 10.7 ms
// (matrix pipe busy 46 -> 55 %, 39 % operand generation on the VALU, ~5 % idle).
// ---------------------------------------------------------------------------------------
// MIRROR PAIRS (round 5).  Operand generation is the part of both kernels that nothing hides, and its count was at
// its minimum -- one phasor per (antenna, pixel, channel, time).  Arrays with point symmetry (hexagonal cores, grids, rings:
// the layouts simulations are run on) have more structure: for two antennas with r' - c = -(r - c) the phasors are complex
// conjugates, E' = conj(E), whatever the direction and the channel.  The host (ops._mirror_order) finds such pairs, measures the
// block's positions from their common centre c (visibilities only see position differences) and orders the rows so that rows
// 16 g + 8 + i hold the mirror antennas of rows 16 g + i (i < 8) for the 16-row groups g named in `mirror`; the kernels then
// evaluate the first octet and write / use its conjugate for the second: the weighted f16 hi / lo halves of the real plane as
// they are, those of the imaginary plane with the sign bits flipped (the split rounds toward zero: exact).  Groups without the
// bit, and antennas without a partner (placed in such groups), are evaluated as before.  Forward: the sweeps of a wave pair
// now walk the octets of ONE 16-row group (rows 32 (u >> 1) + 16 p + 8 (u & 1) + i instead of 16 u + 8 p + i: the same octets,
// dealt differently), so that an odd sweep is the conjugate of the sweep before it in the same lanes.  Backward: the two
// octets of a K step are the jq = 0 / 1 halves of the fragment a lane generates.  Measured, mirror pairs on / off on one box:
// C4 (127-antenna hexagon + outrigger: 7 of 8 groups) 85.4 -> 79.4 ms/step, C3 20.3 -> 18.1, C2 0.861 -> 0.825
// (profiles/r05/mirror_pairs.txt; a timing-only build had promised twice as much for the forward -- an artefact: its
// duplicated image rows lowered the matrix pipe's switching power, and the chip is power-limited under these kernels).
constexpr int MF_KP = 32;                       // pixels per panel (one barrier per panel); 16 per MFMA
constexpr int MF_NH = MF_KP / 16;               // 16-pixel K steps per panel
constexpr int MF_ROWB = 4 * MF_KP + 16;         // [re KP x f16][im KP x f16][pad]: odd number of 16-B granules

// Block shapes.  Diagonal block: one group of <= 128 antennas against itself, upper-triangular
// tiles, 4 waves.  Cross block (arrays with more than 128 antennas): group I (image rows 0..127,
// the sign-carrying L side) against group J (rows 128..255), all 16 tiles, 8 waves, 1 block per CU.
// SELF (complex psky, forward): a diagonal block run as the cross block of a group with ITSELF -- rows 0..32 TI - 1
// hold L = 2^7 E, rows 32 TI.. hold B = psky E of the SAME antennas, both written from one evaluation of E, and only
// the upper-triangular tiles (12 MFMAs each: L and B differ, no diagonal-tile symmetry) are contracted: one pass
// instead of the two real-plane passes of the diagonal kernel (12 TA (TA + 1) / 2 against 2 x (12 TA (TA - 1) / 2
// + 7 TA) MFMAs per K step, half the trigonometry).
template <int TI_, int TJ_, bool CROSS_, bool SELF_ = false>
struct FwdShape {
    static_assert(!SELF_ || (CROSS_ && TI_ == TJ_), "a self block is a cross block of a group with itself");
    static constexpr int TA = TI_;                                 // diagonal block: tiles per side
    static constexpr int TI = TI_, TJ = CROSS_ ? TJ_ : TI_;        // cross block: row tiles (group I) x column tiles (group J)
    static constexpr bool CROSS = CROSS_, SELF = SELF_;
    static constexpr int NT = SELF ? TI * (TI + 1) / 2 : (CROSS ? TI * TJ : TA * (TA + 1) / 2);   // 32x32 output tiles
    static constexpr int NU = 2 * NT;
    // a single tile has two (re | im) units for four waves: the two K steps of a panel go to different
    // waves and the partial tiles are added in the epilogue
    static constexpr bool KSPLIT = NT == 1;
    static constexpr int NW = (CROSS && NT >= 8) ? 8 : 4;                     // waves per block (2-wave blocks for <= 64
                                                                   // antennas: faster or slower with the grid size)
    static constexpr int ROWS = CROSS ? 32 * (TI + TJ) : 32 * TA;  // antenna rows of the LDS images (smaller
                                                                   // arrays: more blocks per CU, 5-7 % faster)
    static constexpr int GROWS = NW * 8;                           // antenna rows per generation sweep
    static constexpr int GEN = ROWS / GROWS;                       // antennas per thread and half panel
    static constexpr int GEN_I = CROSS ? 32 * TI / GROWS : 0;      // sweeps that belong to group I (cross blocks)
    static constexpr int IMG = ROWS * MF_ROWB;                     // one image (hi or lo)
    static constexpr int BUF = 2 * IMG + 64;                       // hi + lo + sign dwords of the panel
    // epilogue scratch: four 32 x 33 float transposition tiles (+ two 16 x 64 float exchange tiles when K is split)
    static constexpr size_t EPI = 4 * 33 * 32 * 4 + (KSPLIT ? 2 * 16 * 64 * 4 : 0);
    static constexpr size_t LDS = 2 * (size_t)BUF < EPI ? EPI : 2 * (size_t)BUF;
    static constexpr int UPW = KSPLIT ? 1 : (NU + NW - 1) / NW;
    static_assert(ROWS % GROWS == 0 && (!CROSS || (32 * TI) % GROWS == 0), "generation sweeps must tile the image rows");
};
__host__ __device__ constexpr int tri_row(int TA, int idx) { int ti = 0; while (idx >= TA - ti) { idx -= TA - ti; ++ti; } return ti; }
__host__ __device__ constexpr int tri_col(int TA, int idx) { int ti = 0; while (idx >= TA - ti) { idx -= TA - ti; ++ti; } return ti + idx; }
// 128 antennas: the ten upper-triangular tiles in the order (0,0) (0,1) (0,2) (0,3) (3,3) (1,1) (1,2)
// (1,3) (2,2) (2,3), so that the four waves' unit ranges touch 1, 2, 1, 2 row tiles (row-major
// order: 1, 2, 2, 2) -- every row tile a wave touches costs 16 sign-mask v_xor per K step
__host__ __device__ constexpr int tri4_row(int t) { return t < 4 ? 0 : (t == 4 ? 3 : (t < 8 ? 1 : 2)); }
__host__ __device__ constexpr int tri4_col(int t) { return t < 4 ? t : (t == 4 ? 3 : (t < 8 ? t - 4 : t - 6)); }
// 33..64 antennas: the three tiles in the order (0,1) (0,0) (1,1) and the unit ranges [0,1) [1,2) [2,4)
// [4,6): 6, 6, 7, 7 MFMAs per wave and K step (two units per wave in row-major order: 7, 12, 7, 0)
template <class SH> __host__ __device__ constexpr int tile_row(int t)
{
    if (SH::SELF) return tri_row(SH::TI, t);
    return SH::CROSS ? t / SH::TJ : (SH::TA == 4 ? tri4_row(t) : (SH::TA == 2 ? (t == 2 ? 1 : 0) : tri_row(SH::TA, t)));
}
template <class SH> __host__ __device__ constexpr int tile_col(int t)
{
    // 65..96 antennas: (0,1) (0,0) (0,2) (1,1) (1,2) (2,2) -- off-diagonal and diagonal tiles alternate, so the
    // three-unit ranges cost 16, 15, 13, 13 MFMAs per K step (row-major order: 13, 18, 13, 13)
    if (SH::SELF) return tri_col(SH::TI, t);
    return SH::CROSS ? t % SH::TJ : (SH::TA == 4 ? tri4_col(t) : (SH::TA == 2 ? (t == 1 ? 0 : 1)
                              : (SH::TA == 3 && t < 2 ? 1 - t : tri_col(SH::TA, t))));
}
// units [unit_begin(w), unit_begin(w + 1)) belong to wave w
template <class SH> __host__ __device__ constexpr int unit_begin(int w)
{
    if (SH::KSPLIT) return w >> 1;                 // waves (0, 1): unit 0, waves (2, 3): unit 1; K step = w & 1
    if (SH::SELF && SH::TI == 4) return w < 4 ? 3 * w : 12 + 2 * (w - 4);      // 20 units over 8 waves: 3 3 3 3 2 2 2 2
    if (SH::SELF && SH::TI == 2) return w < 2 ? 2 * w : (w == 2 ? 4 : (w == 3 ? 5 : 6));     // 6 units: 2 2 1 1
    if (!SH::CROSS && SH::TA == 2) return w <= 2 ? w : (w == 3 ? 4 : 6);
    return SH::UPW * w < SH::NU ? SH::UPW * w : SH::NU;
}

template <class SH> __host__ __device__ constexpr bool is_diag(int t) { return !SH::CROSS && tile_row<SH>(t) == tile_col<SH>(t); }

// x / 2 on the eight f16 values of a fragment (exact but for subnormal results)
__device__ __forceinline__ uint4 half_frag(const uint4& v)
{
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 hf = {(_Float16)0.5f, (_Float16)0.5f};
    uint4 r;
    r.x = __builtin_bit_cast(uint32_t, __builtin_bit_cast(h2, v.x) * hf);
    r.y = __builtin_bit_cast(uint32_t, __builtin_bit_cast(h2, v.y) * hf);
    r.z = __builtin_bit_cast(uint32_t, __builtin_bit_cast(h2, v.z) * hf);
    r.w = __builtin_bit_cast(uint32_t, __builtin_bit_cast(h2, v.w) * hf);
    return r;
}

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

#define RIME_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(as_frag(a), as_frag(b), c, 0, 0, 0)

// CPLX (cross blocks only): psky is complex and the block contracts it in ONE pass with asymmetric
// weighting -- group I rows hold L = 2^7 E, group J rows hold B = (psky scale / 2^7) E (a complex product,
// two more FMAs per generated value), so V = L^H B needs the same MFMAs as one real plane and no sign
// masks.  (Diagonal blocks would need two images of the same antennas; they take the two real passes.)
// MIR: the block has mirror groups (AntArgs.mirror != 0): a kernel instantiation of its own, so that the one which serves arrays
// without symmetry carries no test.  An odd sweep of a mirror group stores the conjugate of the sweep before it (wave-uniform
// test of the mask); every sweep's coordinates stay in registers.  Three leaner forms were measured on one box and lost
// (profiles/r05/mirror_pairs.txt): odd-sweep coordinates fetched from memory on demand (the compiler then drains the panel
// prefetch at the branch: the whole gain gone), from an LDS table (- 5.3 % against - 5.7 %), and a two-phase form with ONE
// rolled copy of the plain odd sweeps (- 3.6 %).
template <class SH, int W, bool SIGNED, bool CPLX, bool MIR = false>
__device__ __forceinline__ void ant_fwd_body(const AntArgs& A, unsigned char* smem)
{
    static_assert(!CPLX || (SH::CROSS && !SIGNED), "complex single pass: cross blocks, no sign masks");
    constexpr int UPW = SH::UPW, U0 = unit_begin<SH>(W);                                // this wave's units: [U0, UE)
    constexpr int UE = SH::KSPLIT ? U0 + 1 : unit_begin<SH>(W + 1);
    constexpr int MF_IMG = SH::IMG, MF_BUF = SH::BUF;
    constexpr int BROW = SH::CROSS ? SH::TI : 0;     // image row-tile offset of the column (B) side
    const int tid = threadIdx.x, lane = tid & 63;
    // 1-D grid, channel fastest: blocks that run together share (t, split), i.e. the same pointing
    // vectors (L2 hits), and no grid dimension hits the 65535 cap
    const int f = __builtin_amdgcn_readfirstlane(blockIdx.x % A.Nf), ts = blockIdx.x / A.Nf;
    const int t = __builtin_amdgcn_readfirstlane(ts / A.S), split = __builtin_amdgcn_readfirstlane(ts % A.S);

    const double nu_c = A.sign * A.freqs[f] * (1.0 / 2.99792458e8);
    const float scl = A.scale[t * A.Nf + f];
    const float* arow = A.psky + (size_t)t * A.st_t + (size_t)f * A.st_f;
    const double* sd = A.sdir + (size_t)t * 3 * A.Pstride;
    const int st_p = __builtin_amdgcn_readfirstlane((int)A.st_p);

    // generation mapping: lane = (pixel pair pp, antenna slot ag); rows of one ds_write are 2 apart
    // (80-B rows: 8 rows x 32 B land in 16 distinct 16-B granules of the 64 banks)
    const int pp = lane & 7, ag = lane >> 3;
    // rows of one sweep: 2 apart inside a wave (bank-conflict-free writes), waves interleaved
    const int grow = SH::NW == 8 ? 2 * ag + 16 * (W & 3) + (W >> 2)
                   : SH::NW == 4 ? 2 * ag + 16 * (W & 1) + (W >> 1) : 2 * ag + W;
    // 33..64 antennas (OCT): a wave generates ONE 16-pixel half of the panel (W & 1) for the rows
    // 16 k + (W >> 1) + 2 ag, k = 0..3, and skips the k whose 8 rows are all padding: 37 antennas cost
    // 3 sweeps per wave instead of 4 (padding rows of the images are never written: they only reach
    // the result rows / columns of padding antennas, which have no baseline slot)
    // OCT8: the 8-wave real-psky cross blocks (arrays of more than 128 antennas, 1-pol): sweeps of 32 rows as OCTX below
    constexpr bool OCT8 = SH::CROSS && !SH::SELF && !CPLX && SH::NW == 8;
    // (three-row-tile blocks keep the two-half mapping -- + 0.5 % with this one -- unless they have mirror groups to gain from it)
    constexpr bool OCT = OCT8 || (!SH::CROSS && (SH::TA != 3 || MIR));
    // the same idea for the 8-wave complex-psky blocks (128 x 128 cross blocks, 128-antenna self blocks; C5): a wave generates
    // one half of the panel for twice as many antennas per lane -- sweeps of 32 rows, rows 32 u + 2 ag + 16 ((W >> 1) & 1) + (W >> 2)
    constexpr bool OCTX = SH::CROSS && CPLX;
    constexpr int SWX = SH::NW * 4;                    // rows of an OCTX sweep: 32 (8 waves) or 16 (4 waves)
    constexpr int NGEN = OCT8 ? SH::ROWS / 32 : OCT ? SH::ROWS / 16 : (OCTX ? 2 : 1) * (SH::SELF ? SH::GEN_I : SH::GEN);
    // (round 4) the rows of a sweep are one OCTET per wave pair -- rows 16 u + 8 (W >> 1) + 2 (ag & 3) + (ag >> 2), conflict-free
    // ds_write_b32 as before -- instead of the rows of one parity: the sweeps a wave skips are then whole octets of padding,
    // 19 antennas cost 2 + 1 sweeps per wave pair instead of 2 + 2, 37 antennas 3 + 2 instead of 3 + 3 (same bits)
    // (round 5) ... and the sweeps of a wave pair p = W >> 1 walk the octets of ONE 16-row group after the other: sweep u = rows
    // 32 (u >> 1) + 16 p + 8 (u & 1) + i (before: 16 u + 8 p + i -- the same octets, dealt differently: same bits), so that
    // the mirror antennas of a sweep's rows (AntArgs.mirror) are the next sweep's rows of the same lanes
    const int orow = 16 * (W >> 1) + 2 * (ag & 3) + (ag >> 2);
    auto octet_row = [&](int u) { return 32 * (u >> 1) + 8 * (u & 1) + orow; };
    // sweeps whose octet starts below Nant (monotone in u): uniform
    const int mrow = A.Nant - 16 * (W >> 1);
    const int nk = (OCT && !OCT8) ? min(NGEN, (max(mrow, 0) + 31) / 32 + (max(mrow - 8, 0) + 31) / 32) : NGEN;
    const int growx = SH::NW == 8 ? 2 * ag + 16 * ((W >> 1) & 1) + (W >> 2) : 2 * ag + (W >> 1);
    static_assert(!MIR || (OCT && !OCT8), "mirror groups: diagonal blocks with the half-panel generation mapping");
    double ax[NGEN], ay[NGEN], az[NGEN];
#pragma unroll
    for (int u = 0; u < NGEN; ++u) {
        const int an = OCT8 ? 32 * u + growx : OCT ? octet_row(u) : (OCTX ? SWX * u + growx : SH::GROWS * u + grow);
        const bool ok = an < A.Nant;
        ax[u] = ok ? nu_c * A.antpos[3 * an] : 0.0;
        ay[u] = ok ? nu_c * A.antpos[3 * an + 1] : 0.0;
        az[u] = ok ? nu_c * A.antpos[3 * an + 2] : 0.0;
    }
    const int goff = grow * MF_ROWB + pp * 4;

    f32x16 acc[UPW][2];
#pragma unroll
    for (int s = 0; s < UPW; ++s)
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc[s][0][e] = 0.f; acc[s][1][e] = 0.f; }

    const int npanel = A.Pstride / MF_KP;
    const int pbeg = __builtin_amdgcn_readfirstlane(split * A.panels_per_split);
    const int pend = __builtin_amdgcn_readfirstlane(min(npanel, pbeg + A.panels_per_split));
    if (pbeg >= pend) return;                        // uniform over the block

    // panel fetch: wave-uniform bases (scalar address arithmetic) + constant 32-bit lane offsets;
    // psky is read as two dwords with the runtime pixel stride (a branch on the stride makes the
    // compiler issue both variants with 64-bit vector multiplies: 17 % of the loop's VALU work)
    double2 sx[MF_NH], sy[MF_NH], sz[MF_NH]; float2 av[MF_NH]; float2 aw[CPLX ? MF_NH : 1];   // aw: imaginary plane
    const uint32_t lo_s = 16u * pp, lo_a0 = 8u * pp * (uint32_t)st_p, lo_a1 = lo_a0 + 4u * (uint32_t)st_p;
    const double* sdy = sd + A.Pstride;
    const double* sdz = sd + 2 * (size_t)A.Pstride;
    // buffer loads: descriptor (SGPRs, wave-uniform: built from kernel arguments and the block index) + constant per-lane
    // offset + scalar panel offset -- no vector address arithmetic in the generation phase, whose cost is its
    // instruction count (profiles/r03/lab_forward_experiments.txt); out-of-range reads return 0 instead of faulting
    // (the descriptor inputs go through readfirstlane: a base pointer the compiler cannot PROVE wave-uniform makes it wrap
    // every buffer load in a waterfall loop of ~14 instructions -- seen here on the psky descriptor, guide T20)
    auto uniform_ptr = [](const void* q) {
        const unsigned long long a = reinterpret_cast<unsigned long long>(q);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
        return reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo);
    };
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        uniform_ptr(sd), 0, __builtin_amdgcn_readfirstlane((int)min((long long)3 * A.Pstride * 8, 0x7fffffffLL)), 0x00020000);
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        uniform_ptr(arow), 0, __builtin_amdgcn_readfirstlane((int)min((long long)A.Pstride * st_p * 4, 0x7fffffffLL)), 0x00020000);
    (void)sdy; (void)sdz;
    auto fetch = [&](int panel, int hf) {
        const int p0 = panel * MF_KP + 16 * hf;      // uniform
        sx[hf] = __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)lo_s, p0 * 8, 0));
        sy[hf] = __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)lo_s, (A.Pstride + p0) * 8, 0));
        sz[hf] = __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)lo_s, (2 * A.Pstride + p0) * 8, 0));
        const int so = p0 * st_p * 4;
        av[hf] = make_float2(__uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ra, (int)lo_a0, so, 0)),
                             __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ra, (int)lo_a1, so, 0)));
        if constexpr (CPLX)
            aw[hf] = make_float2(__uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ra, (int)lo_a0 + 4, so, 0)),
                                 __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ra, (int)lo_a1 + 4, so, 0)));
    };
    auto generate = [&](unsigned char* buf, int next_panel) {
        if constexpr (OCT) {
            constexpr int hf = W & 1;
            const float w0 = __builtin_amdgcn_sqrtf(fabsf(av[hf].x) * scl), w1 = __builtin_amdgcn_sqrtf(fabsf(av[hf].y) * scl);
            if (SIGNED && W < 2 && lane < 8)
                *reinterpret_cast<uint32_t*>(buf + 2 * MF_IMG + 4 * (8 * hf + pp)) =
                    ((__float_as_uint(av[hf].x) >> 16) & 0x8000u) | (__float_as_uint(av[hf].y) & 0x80000000u);
            // one evaluated sweep: the weighted f16 halves of this lane's pixel pair for the antenna at (bx, by, bz)
            auto sweep = [&](double bx, double by, double bz, uint32_t& rh, uint32_t& rl, uint32_t& ih, uint32_t& il) {
                const double ph0 = phase3(bx, sx[hf].x, by, sy[hf].x, bz, sz[hf].x);
                const double ph1 = phase3(bx, sx[hf].y, by, sy[hf].y, bz, sz[hf].y);
                const float r0 = turn_frac(ph0), r1 = turn_frac(ph1);
                const float s0 = __builtin_amdgcn_sinf(r0), c0 = __builtin_amdgcn_cosf(r0);
                const float s1 = __builtin_amdgcn_sinf(r1), c1 = __builtin_amdgcn_cosf(r1);
                split2(w0 * c0, w1 * c1, rh, rl);
                split2(w0 * s0, w1 * s1, ih, il);
            };
            auto store = [&](int row, uint32_t rh, uint32_t rl, uint32_t ih, uint32_t il) {
                unsigned char* o = buf + row * MF_ROWB + pp * 4 + 32 * hf;
                *reinterpret_cast<uint32_t*>(o) = rh;
                *reinterpret_cast<uint32_t*>(o + 2 * MF_KP) = ih;
                *reinterpret_cast<uint32_t*>(o + MF_IMG) = rl;
                *reinterpret_cast<uint32_t*>(o + MF_IMG + 2 * MF_KP) = il;
            };
            if constexpr (MIR) {
                uint32_t m_rh = 0, m_rl = 0, m_ih = 0, m_il = 0;
#pragma unroll
                for (int u = 0; u < NGEN; ++u) {
                    if (u < nk) {
                        uint32_t rh, rl, ih, il;
                        if ((u & 1) && ((A.mirror >> (2 * (u >> 1) + (W >> 1))) & 1)) {
                            rh = m_rh; rl = m_rl; ih = m_ih ^ 0x80008000u; il = m_il ^ 0x80008000u;
                        } else sweep(ax[u], ay[u], az[u], rh, rl, ih, il);
                        if (!(u & 1)) { m_rh = rh; m_rl = rl; m_ih = ih; m_il = il; }
                        store(octet_row(u), rh, rl, ih, il);
                    }
                }
            } else {
#pragma unroll
                for (int u = 0; u < NGEN; ++u) {
                    if (u < nk) {
                        uint32_t rh, rl, ih, il;
                        sweep(ax[u], ay[u], az[u], rh, rl, ih, il);
                        store(OCT8 ? 32 * u + growx : octet_row(u), rh, rl, ih, il);
                    }
                }
            }
            fetch(next_panel, hf);
            return;
        }
        if constexpr (CPLX) {
            const float sb = scl * (1.0f / 128.0f), si = sb * A.imsign;
            constexpr int HF0 = OCTX ? (W & 1) : 0, HF1 = OCTX ? (W & 1) + 1 : MF_NH;
            constexpr int NI = (OCTX ? 2 : 1) * SH::GEN_I;      // sweeps that belong to group I
#pragma unroll
            for (int hf = HF0; hf < HF1; ++hf) {
                const float ar0 = av[hf].x * sb, ar1 = av[hf].y * sb, ai0 = aw[hf].x * si, ai1 = aw[hf].y * si;
#pragma unroll
                for (int u = 0; u < NGEN; ++u) {
                    const double ph0 = phase3(ax[u], sx[hf].x, ay[u], sy[hf].x, az[u], sz[hf].x);
                    const double ph1 = phase3(ax[u], sx[hf].y, ay[u], sy[hf].y, az[u], sz[hf].y);
                    const float r0 = turn_frac(ph0), r1 = turn_frac(ph1);
                    const float s0 = __builtin_amdgcn_sinf(r0), c0 = __builtin_amdgcn_cosf(r0);
                    const float s1 = __builtin_amdgcn_sinf(r1), c1 = __builtin_amdgcn_cosf(r1);
                    uint32_t rh, rl, ih, il;
                    unsigned char* o = OCTX ? buf + (SWX * u + growx) * MF_ROWB + pp * 4 + 32 * hf
                                            : buf + goff + u * SH::GROWS * MF_ROWB + 32 * hf;
                    if (SH::SELF || u < NI) { // group I: L = 2^7 E
                        split2(128.0f * c0, 128.0f * c1, rh, rl);
                        split2(128.0f * s0, 128.0f * s1, ih, il);
                        *reinterpret_cast<uint32_t*>(o) = rh;
                        *reinterpret_cast<uint32_t*>(o + 2 * MF_KP) = ih;
                        *reinterpret_cast<uint32_t*>(o + MF_IMG) = rl;
                        *reinterpret_cast<uint32_t*>(o + MF_IMG + 2 * MF_KP) = il;
                        if constexpr (SH::SELF) o += SH::GEN_I * SH::GROWS * MF_ROWB;     // the same antenna's B row
                    }
                    if (SH::SELF || u >= NI) { // group J: B = psky E (complex product)
                        split2(fmaf(ar0, c0, -ai0 * s0), fmaf(ar1, c1, -ai1 * s1), rh, rl);
                        split2(fmaf(ar0, s0, ai0 * c0), fmaf(ar1, s1, ai1 * c1), ih, il);
                        *reinterpret_cast<uint32_t*>(o) = rh;
                        *reinterpret_cast<uint32_t*>(o + 2 * MF_KP) = ih;
                        *reinterpret_cast<uint32_t*>(o + MF_IMG) = rl;
                        *reinterpret_cast<uint32_t*>(o + MF_IMG + 2 * MF_KP) = il;
                    }
                }
                fetch(next_panel, hf);
            }
            return;
        }
#pragma unroll
        for (int hf = 0; hf < MF_NH; ++hf) {
            const float w0 = __builtin_amdgcn_sqrtf(fabsf(av[hf].x) * scl), w1 = __builtin_amdgcn_sqrtf(fabsf(av[hf].y) * scl);
            if (SIGNED && tid < 8)
                *reinterpret_cast<uint32_t*>(buf + 2 * MF_IMG + 4 * (8 * hf + pp)) =
                    ((__float_as_uint(av[hf].x) >> 16) & 0x8000u) | (__float_as_uint(av[hf].y) & 0x80000000u);
#pragma unroll
            for (int u = 0; u < SH::GEN; ++u) {
                const double ph0 = phase3(ax[u], sx[hf].x, ay[u], sy[hf].x, az[u], sz[hf].x);
                const double ph1 = phase3(ax[u], sx[hf].y, ay[u], sy[hf].y, az[u], sz[hf].y);
                const float r0 = turn_frac(ph0), r1 = turn_frac(ph1);
                const float s0 = __builtin_amdgcn_sinf(r0), c0 = __builtin_amdgcn_cosf(r0);
                const float s1 = __builtin_amdgcn_sinf(r1), c1 = __builtin_amdgcn_cosf(r1);
                uint32_t rh, rl, ih, il;
                split2(w0 * c0, w1 * c1, rh, rl);
                split2(w0 * s0, w1 * s1, ih, il);
                unsigned char* o = buf + goff + u * SH::GROWS * MF_ROWB + 32 * hf;
                *reinterpret_cast<uint32_t*>(o) = rh;
                *reinterpret_cast<uint32_t*>(o + 2 * MF_KP) = ih;
                *reinterpret_cast<uint32_t*>(o + MF_IMG) = rl;
                *reinterpret_cast<uint32_t*>(o + MF_IMG + 2 * MF_KP) = il;
            }
            fetch(next_panel, hf);                   // latency hidden by the MFMA phase
        }
    };

    const int foff = (lane & 31) * MF_ROWB + (lane >> 5) * 16;   // fragment: row, k-half
    // (a wave-uniform fast path that skips the masks for sign-free panels was tried: the branch
    // around the MFMA block makes the register allocator spill the accumulators -- 4x slower)
    auto contract = [&](const unsigned char* buf) {
        auto frag = [&](int tile, int img, int im, int ks) {
            return *reinterpret_cast<const uint4*>(buf + img * MF_IMG + tile * 32 * MF_ROWB + foff + im * 2 * MF_KP + 32 * ks);
        };
        auto sfrag = [&](int tile, int img, int im, int ks, const uint4& sg) {
            uint4 v = frag(tile, img, im, ks);
            if constexpr (SIGNED) { v.x ^= sg.x; v.y ^= sg.y; v.z ^= sg.z; v.w ^= sg.w; }
            return v;
        };
        constexpr int T0 = U0 >> 1, T1 = (UE + 1) >> 1;               // its tiles: [T0, T1)
        // K step outermost: only one K step's row fragments are live at a time (16 registers fewer)
#pragma unroll
        for (int ks = 0; ks < MF_NH; ++ks) {
            if constexpr (SH::KSPLIT) { if (ks != (W & 1)) continue; }
            uint4 Lrh, Lih, Lrl, Lil;                                 // sign-applied row-tile fragments
            static_for<T0, T1>([&](auto tc) {
                constexpr int tile = decltype(tc)::value;
                constexpr bool hasR = 2 * tile >= U0, hasI = 2 * tile + 1 < UE;
                constexpr int sR = hasR ? 2 * tile - U0 : 0, sI = hasI ? 2 * tile + 1 - U0 : 0;   // accumulator slots
                constexpr int ti = tile_row<SH>(tile), tj = BROW + tile_col<SH>(tile);
                if constexpr (tile == T0 || tile_row<SH>(tile > 0 ? tile - 1 : 0) != ti) {
                    uint4 sg = make_uint4(0, 0, 0, 0);
                    if constexpr (SIGNED) sg = *reinterpret_cast<const uint4*>(buf + 2 * MF_IMG + (2 * ks + (lane >> 5)) * 16);
                    Lrh = sfrag(ti, 0, 0, ks, sg); Lih = sfrag(ti, 0, 1, ks, sg);
                    Lrl = sfrag(ti, 1, 0, ks, sg); Lil = sfrag(ti, 1, 1, ks, sg);
                }
                const uint4 Brh = frag(tj, 0, 0, ks), Bih = frag(tj, 0, 1, ks), Brl = frag(tj, 1, 0, ks), Bil = frag(tj, 1, 1, ks);
                if constexpr (is_diag<SH>(tile)) {
                    // diagonal tile: L and B are the same image rows (up to the pixel sign), so the lo x hi
                    // products are the transposes of the hi x lo ones, and Li.Br is the transpose of Lr.Bi:
                    //   Vr = A + A^T,  A = (Lrh/2).Brh + (Lih/2).Bih + Lrh.Brl + Lih.Bil      (4 MFMAs, not 6)
                    //   Vi = A - A^T,  A = Lrh.Bih + Lrh.Bil - Lih.Brl                        (3 MFMAs, not 6)
                    // the transposes are taken in the epilogue
                    if constexpr (hasR) {
                        const uint4 Hr = half_frag(Lrh), Hi = half_frag(Lih);
                        acc[sR][0] = RIME_MFMA(Hr, Brh, acc[sR][0]);
                        acc[sR][0] = RIME_MFMA(Hi, Bih, acc[sR][0]);
                        acc[sR][0] = RIME_MFMA(Lrh, Brl, acc[sR][0]);
                        acc[sR][0] = RIME_MFMA(Lih, Bil, acc[sR][0]);
                    }
                    if constexpr (hasI) {
                        acc[sI][0] = RIME_MFMA(Lrh, Bih, acc[sI][0]);
                        acc[sI][1] = RIME_MFMA(Lih, Brl, acc[sI][1]);
                        acc[sI][0] = RIME_MFMA(Lrh, Bil, acc[sI][0]);
                    }
                    return;
                }
                // real part Lr.Br + Li.Bi -> acc[sR][0]; imaginary part Lr.Bi -> acc[sI][0], Li.Br -> acc[sI][1]
                if constexpr (hasR) acc[sR][0] = RIME_MFMA(Lrh, Brh, acc[sR][0]);
                if constexpr (hasI) acc[sI][0] = RIME_MFMA(Lrh, Bih, acc[sI][0]);
                if constexpr (hasR) acc[sR][0] = RIME_MFMA(Lih, Bih, acc[sR][0]);
                if constexpr (hasI) acc[sI][1] = RIME_MFMA(Lih, Brh, acc[sI][1]);
                if constexpr (hasR) acc[sR][0] = RIME_MFMA(Lrh, Brl, acc[sR][0]);
                if constexpr (hasI) acc[sI][0] = RIME_MFMA(Lrh, Bil, acc[sI][0]);
                if constexpr (hasR) acc[sR][0] = RIME_MFMA(Lih, Bil, acc[sR][0]);
                if constexpr (hasI) acc[sI][1] = RIME_MFMA(Lih, Brl, acc[sI][1]);
                if constexpr (hasR) acc[sR][0] = RIME_MFMA(Lrl, Brh, acc[sR][0]);
                if constexpr (hasI) acc[sI][0] = RIME_MFMA(Lrl, Bih, acc[sI][0]);
                if constexpr (hasR) acc[sR][0] = RIME_MFMA(Lil, Bih, acc[sR][0]);
                if constexpr (hasI) acc[sI][1] = RIME_MFMA(Lil, Brh, acc[sI][1]);
            });
        }
    };
    // two panels per trip: buffer addresses are compile-time offsets
    unsigned char* const buf0 = smem;
    unsigned char* const buf1 = smem + MF_BUF;
    if constexpr (OCT || OCTX) fetch(pbeg, W & 1);
    else {
#pragma unroll
        for (int hf = 0; hf < MF_NH; ++hf) fetch(pbeg, hf);
    }
    generate(buf0, min(pbeg + 1, pend - 1));
    __syncthreads();
    for (int panel = pbeg; panel < pend; panel += 2) {
        if (panel + 1 < pend) generate(buf1, min(panel + 2, pend - 1));
        contract(buf0);
        __syncthreads();
        if (panel + 1 < pend) {
            if (panel + 2 < pend) generate(buf0, min(panel + 3, pend - 1));
            contract(buf1);
        }
        __syncthreads();
    }

    // epilogue: V[i,j] / scale -> the baseline slot(s) of pair (i,j) of this block's slab
    // ws[split][t][f][re|im][Nbl]: consecutive lanes (columns j) hit consecutive baseline slots, so
    // the stores coalesce (a scattered store into the [Nbl][Nt][Nf][2] result costs a 32-B sector per
    // 4-B value: 8 GB instead of 1.6 GB per launch at C4); reduce_vis_kernel sums the splits in a
    // fixed order and transposes into the result layout.
    float* dst = A.ws + (((size_t)split * A.Nt + t) * A.Nf + f) * 2 * A.Nbl;
    const float inv = 1.0f / scl;
    // (the three-row-tile self block is one register over its budget here: its epilogue re-derives the lane index from the
    //  execution mask instead of keeping `lane & 31` / `lane >> 5` alive across the panel loop -- until round 5 that was 8 bytes
    //  of scratch, a store before the loop and a load after it)
    int elane = lane;
    if constexpr (SH::SELF && SH::TI == 3) elane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int col = elane & 31;
    RIME_MFMA_SETTLE();
    // the image buffers are free after the loop's last barrier: a private 32 x 33 float tile per wave
    // transposes the accumulators of diagonal-tile units (written and read by this wave only)
    float* tr = reinterpret_cast<float*>(smem) + W * (32 * 33);
#pragma unroll
    for (int s = 0; s < UPW; ++s) {
        const int u = U0 + s;
        if (u < UE) {
            const int ti = tile_row<SH>(u >> 1), tj = tile_col<SH>(u >> 1), im = u & 1;
            f32x16 val;
#pragma unroll
            for (int e = 0; e < 16; ++e) val[e] = im ? acc[s][0][e] - acc[s][1][e] : acc[s][0][e];
            if constexpr (SH::KSPLIT) {
                // the odd wave of a pair hands its K step's partial tile to the even one ([e][lane] floats
                // behind the four transposition tiles)
                float* ex = reinterpret_cast<float*>(smem) + 4 * (32 * 33) + (W >> 1) * (16 * 64);
                if constexpr ((W & 1) == 1) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) ex[e * 64 + lane] = val[e];
                }
                __syncthreads();
                if constexpr ((W & 1) == 1) return;
#pragma unroll
                for (int e = 0; e < 16; ++e) val[e] += ex[e * 64 + lane];
            }
            if (is_diag<SH>(u >> 1)) {
#pragma unroll
                for (int e = 0; e < 16; ++e) tr[((e & 3) + 8 * (e >> 2) + 4 * (elane >> 5)) * 33 + col] = val[e];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float tv = tr[col * 33 + (e & 3) + 8 * (e >> 2) + 4 * (elane >> 5)];
                    val[e] = im ? val[e] - tv : val[e] + tv;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (e & 3) + 8 * (e >> 2) + 4 * (elane >> 5);
                const int i = ti * 32 + row, j = tj * 32 + col;      // indices inside group I / group J
                const float v = val[e] * inv;
                const int bd = A.pair_direct[i * MF_NA + j];
                if (bd >= 0) dst[(size_t)im * A.Nbl + bd] = v;
                const int bc = A.pair_conj[i * MF_NA + j];
                if (bc >= 0) dst[(size_t)im * A.Nbl + bc] = im ? -v : v;
            }
        }
    }
}

// A block works on one psky row (t, f): rows without a negative value (beam-weighted emission is
// non-negative) take the kernel instantiation without sign masks.  Both instantiations are launched
// over the full grid and a block returns at once when its row belongs to the other one: a branch
// inside one kernel costs registers (the allocator serves the union of both paths and spills).
__device__ __forceinline__ bool row_is_signed(const AntArgs& A)
{
    if (!A.rowmin) return true;
    const int f = blockIdx.x % A.Nf, t = (blockIdx.x / A.Nf) / A.S;
    return A.rowmin[t * A.Nf + f] < 0.f;
}

template <class SH, bool SIGNED, bool CPLX, bool MIR = false>
__device__ __forceinline__ void ant_fwd_dispatch(const AntArgs& A, unsigned char* smem)
{
    if constexpr (!CPLX) { if (row_is_signed(A) != SIGNED) return; }        // uniform over the block
    switch (threadIdx.x >> 6) {                      // wave-uniform: every wave runs the same barriers
        case 0: ant_fwd_body<SH, 0, SIGNED, CPLX, MIR>(A, smem); break;
        case 1: ant_fwd_body<SH, 1, SIGNED, CPLX, MIR>(A, smem); break;
        case 2: if constexpr (SH::NW > 2) ant_fwd_body<SH, 2, SIGNED, CPLX, MIR>(A, smem); break;
        case 3: if constexpr (SH::NW > 2) ant_fwd_body<SH, 3, SIGNED, CPLX, MIR>(A, smem); break;
        case 4: if constexpr (SH::NW > 4) ant_fwd_body<SH, 4, SIGNED, CPLX, MIR>(A, smem); break;
        case 5: if constexpr (SH::NW > 4) ant_fwd_body<SH, 5, SIGNED, CPLX, MIR>(A, smem); break;
        case 6: if constexpr (SH::NW > 4) ant_fwd_body<SH, 6, SIGNED, CPLX, MIR>(A, smem); break;
        default: if constexpr (SH::NW > 4) ant_fwd_body<SH, 7, SIGNED, CPLX, MIR>(A, smem); break;
    }
}

template <int TA> constexpr int fwd_threads() { return FwdShape<TA, TA, false>::NW * 64; }

template <int TA, bool SIGNED, bool MIR = false>
__global__ void __launch_bounds__(fwd_threads<TA>(), 2)
fringe_ant_fwd_kernel(AntArgs A)
{
    extern __shared__ __align__(16) unsigned char smem[];
    ant_fwd_dispatch<FwdShape<TA, TA, false>, SIGNED, false, MIR>(A, smem);
}

// cross blocks: TI x TJ tiles, (1,1) (1,2) (2,2) (4,4); 8 waves for the 16-tile shape
template <int TI, int TJ> constexpr int cross_threads() { return FwdShape<TI, TJ, true>::NW * 64; }
template <int TI, int TJ> constexpr int cross_minwaves() { return FwdShape<TI, TJ, true>::NW == 8 ? 1 : 2; }

template <int TI, int TJ, bool SIGNED, bool CPLX>
__global__ void __launch_bounds__((cross_threads<TI, TJ>()), (cross_minwaves<TI, TJ>()))
fringe_ant_fwd_cross_kernel(AntArgs A)
{
    extern __shared__ __align__(16) unsigned char smem[];
    ant_fwd_dispatch<FwdShape<TI, TJ, true>, SIGNED, CPLX>(A, smem);
}

// self blocks (complex psky): the diagonal block of TI x 32 antennas as a triangular cross block
template <int TI> constexpr int self_threads() { return FwdShape<TI, TI, true, true>::NW * 64; }
template <int TI> constexpr int self_minwaves() { return FwdShape<TI, TI, true, true>::NW == 8 ? 1 : 2; }

template <int TI>
__global__ void __launch_bounds__((self_threads<TI>()), (self_minwaves<TI>()))
fringe_ant_fwd_self_kernel(AntArgs A)
{
    extern __shared__ __align__(16) unsigned char smem[];
    ant_fwd_dispatch<FwdShape<TI, TI, true, true>, false, true>(A, smem);
}


// ---------------------------------------------------------------------------------------
// forward kernel, 33..48 antennas (one diagonal block, real psky): PACKED second row tile  (round 4)
//
// 37 antennas (HERA-37, BASELINE configs[2]) in 64 image rows: the second row tile holds 5 real antennas, and the two
// tiles that touch it -- (0,1) and (1,1) -- cost 12 + 7 of the 26 MFMAs per K step of the generic two-tile shape.  The
// matrix pipe, not the 48 generated rows, is what a block waits for there (profiles/r03/lab_forward_experiments.txt (j)).
// With <= 16 antennas in the second row tile its re and im planes fit into ONE 32-row / 32-column operand:
//     P  = [ Br | Bi ]     (columns 0..15: Br of antennas 32..47, columns 16..31: Bi of the same antennas)
//     Q  = [ Bi | -Br ]
//   tile (0,1):  Lr.P + Li.Q = [ Lr.Br + Li.Bi | Lr.Bi - Li.Br ] = [ Vr | Vi ]            6 MFMAs per K step, not 12,
//                                                                                         ONE accumulator, not three
//   tile (1,1):  [Lr; Li].P  = [[Lr.Br, Lr.Bi], [Li.Br, Li.Bi]]                            3 MFMAs per K step, not 7;
//                Vr = D[a][b] + D[16+a][16+b],  Vi = D[a][16+b] - D[16+a][b]: rows a and 16 + a sit in the same lane
//                (accumulator elements e and e + 8), columns b and 16 + b in lanes 16 apart -- one cross-lane read per
//                element in the epilogue
//   tile (0,0):  the symmetric form of the generic kernel, 7 MFMAs.
// 16 MFMAs per K step instead of 26 (SQ_INSTS_MFMA per launch: Nt Nf (P/16) 16), same three hi/lo split products, same
// f32 accumulation, fixed order.  The second row tile's image holds three planes of 16 "virtual rows" of ONE plane each
// (re, im, -re; 80-byte rows: fragment reads and the generation's ds_write_b32 are bank-conflict free); the first row
// tile keeps the [re | im] rows of the generic kernel.  Deal: each wave takes ONE K step of every panel (w & 1) for one
// tile group -- waves 0, 1: tile (0,0), 7 MFMAs; waves 2, 3: tiles (0,1) + (1,1), 9 MFMAs -- and the two waves of a pair
// exchange one partial tile through LDS in the epilogue, so that every wave finishes and stores one unit.
// Generation: the half-panel mapping of the two-tile blocks (a wave writes ONE 16-pixel half for one octet of rows per sweep).
// ---------------------------------------------------------------------------------------
struct PK {
    static constexpr int NW = 4;
    static constexpr int ROWB0 = MF_ROWB;               // antennas 0..31: [32 px re | 32 px im | pad], 144 B
    static constexpr int IMG0 = 32 * ROWB0;
    static constexpr int ROWB1 = 2 * MF_KP + 16;        // antennas 32..47: 32 px of ONE plane + pad, 80 B (5 slots of 16 B: odd)
    static constexpr int IMG1 = 48 * ROWB1;             // virtual rows 0..15 re, 16..31 im, 32..47 -re
    static constexpr int IMG = IMG0 + IMG1;             // one image (hi or lo)
    static constexpr int BUF = 2 * IMG + 64;            // hi + lo + sign dwords of the panel
    static constexpr size_t EPI = 4 * 33 * 32 * 4 + 5 * 16 * 64 * 4;      // 4 transposition tiles + 5 exchange tiles
    static constexpr size_t LDS = 2 * (size_t)BUF < EPI ? EPI : 2 * (size_t)BUF;
};
static_assert(MF_KP == 32, "the packed shape assumes 32-pixel panels (two K steps, one per wave of a pair)");

template <int W, bool SIGNED, bool MIR = false>
__device__ __forceinline__ void ant_fwd_packed_body(const AntArgs& A, unsigned char* smem)
{
    constexpr int KS = W & 1;                           // this wave's K step of every panel
    constexpr bool PAIR_B = W >= 2;                     // waves 2, 3: tiles (0,1) + (1,1); waves 0, 1: tile (0,0)
    const int tid = threadIdx.x, lane = tid & 63;
    const int f = __builtin_amdgcn_readfirstlane(blockIdx.x % A.Nf), ts = blockIdx.x / A.Nf;
    const int t = __builtin_amdgcn_readfirstlane(ts / A.S), split = __builtin_amdgcn_readfirstlane(ts % A.S);

    const double nu_c = A.sign * A.freqs[f] * (1.0 / 2.99792458e8);
    const float scl = A.scale[t * A.Nf + f];
    const float* arow = A.psky + (size_t)t * A.st_t + (size_t)f * A.st_f;
    const double* sd = A.sdir + (size_t)t * 3 * A.Pstride;
    const int st_p = __builtin_amdgcn_readfirstlane((int)A.st_p);

    // generation: lane = (pixel pair pp, antenna slot ag); this wave writes the 16-pixel half `hf` of the panel for the
    // antennas 16 u + 8 (W >> 1) + 2 (ag & 3) + (ag >> 2), u = 0, 1 (first row tile), 2 (second); sweeps whose octet is all
    // padding are skipped (the waves with fewer sweeps are the ones with more MFMAs)
    const int pp = lane & 7, ag = lane >> 3;
    constexpr int hf = W & 1;
    constexpr int NGEN = 3;
    // (round 5) first row tile: the two sweeps of a wave pair p are the two octets of ONE 16-row group (rows 16 p + 8 u + i; before:
    // 16 u + 8 p + i), so that a mirror group's second octet is the conjugate of the sweep before it in the same lanes
    // (AntArgs.mirror bit p); second row tile: rows 32 + 8 p + i as before (no mirror pairs there)
    const int oct = 2 * (ag & 3) + (ag >> 2);                     // row inside an octet
    auto sweep_row = [&](int u) { return u < 2 ? 16 * (W >> 1) + 8 * u + oct : 32 + 8 * (W >> 1) + oct; };
    const int grow = 8 * (W >> 1) + oct;                          // row inside the second row tile's sweep of 16
    // sweeps whose octet starts below Nant (monotone in u; uniform): 37 antennas -> 3 sweeps (waves 0, 1), 2 (waves 2, 3)
    const int nk = (16 * (W >> 1) < A.Nant) + (16 * (W >> 1) + 8 < A.Nant) + (32 + 8 * (W >> 1) < A.Nant);
    double ax[NGEN], ay[NGEN], az[NGEN];
#pragma unroll
    for (int u = 0; u < NGEN; ++u) {
        const int an = sweep_row(u);
        const bool ok = an < A.Nant;
        ax[u] = ok ? nu_c * A.antpos[3 * an] : 0.0;
        ay[u] = ok ? nu_c * A.antpos[3 * an + 1] : 0.0;
        az[u] = ok ? nu_c * A.antpos[3 * an + 2] : 0.0;
    }

    f32x16 acc0, acc1, acc2, acc3; // waves 0, 1: R (hi x lo), I (two products), R (hi x hi); waves 2, 3: tile (0,1), tile (1,1)
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; acc2[e] = 0.f; acc3[e] = 0.f; }

    const int npanel = A.Pstride / MF_KP;
    const int pbeg = __builtin_amdgcn_readfirstlane(split * A.panels_per_split);
    const int pend = __builtin_amdgcn_readfirstlane(min(npanel, pbeg + A.panels_per_split));
    if (pbeg >= pend) return;                        // uniform over the block

    // panel fetch through buffer loads (see the generic kernel): descriptor in SGPRs + constant lane offset + scalar offset
    double2 sx, sy, sz; float2 av;
    const uint32_t lo_s = 16u * pp, lo_a0 = 8u * pp * (uint32_t)st_p, lo_a1 = lo_a0 + 4u * (uint32_t)st_p;
    auto uniform_ptr = [](const void* q) {
        const unsigned long long a = reinterpret_cast<unsigned long long>(q);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
        return reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo);
    };
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        uniform_ptr(sd), 0, __builtin_amdgcn_readfirstlane((int)min((long long)3 * A.Pstride * 8, 0x7fffffffLL)), 0x00020000);
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        uniform_ptr(arow), 0, __builtin_amdgcn_readfirstlane((int)min((long long)A.Pstride * st_p * 4, 0x7fffffffLL)), 0x00020000);
    auto fetch = [&](int panel) {
        const int p0 = panel * MF_KP + 16 * hf;      // uniform
        sx = __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)lo_s, p0 * 8, 0));
        sy = __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)lo_s, (A.Pstride + p0) * 8, 0));
        sz = __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)lo_s, (2 * A.Pstride + p0) * 8, 0));
        const int so = p0 * st_p * 4;
        av = make_float2(__uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ra, (int)lo_a0, so, 0)),
                         __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ra, (int)lo_a1, so, 0)));
    };
    auto generate = [&](unsigned char* buf, int next_panel) {
        const float w0 = __builtin_amdgcn_sqrtf(fabsf(av.x) * scl), w1 = __builtin_amdgcn_sqrtf(fabsf(av.y) * scl);
        if (SIGNED && W < 2 && lane < 8)
            *reinterpret_cast<uint32_t*>(buf + 2 * PK::IMG + 4 * (8 * hf + pp)) =
                ((__float_as_uint(av.x) >> 16) & 0x8000u) | (__float_as_uint(av.y) & 0x80000000u);
        uint32_t m_rh = 0, m_rl = 0, m_ih = 0, m_il = 0;          // sweep 0: sweep 1 of a mirror group is its conjugate
#pragma unroll
        for (int u = 0; u < NGEN; ++u) {
            if (u < nk) {
                uint32_t rh, rl, ih, il;
                if (MIR && u == 1 && ((A.mirror >> (W >> 1)) & 1)) {   // uniform
                    rh = m_rh; rl = m_rl; ih = m_ih ^ 0x80008000u; il = m_il ^ 0x80008000u;
                } else {
                    const double ph0 = phase3(ax[u], sx.x, ay[u], sy.x, az[u], sz.x);
                    const double ph1 = phase3(ax[u], sx.y, ay[u], sy.y, az[u], sz.y);
                    const float r0 = turn_frac(ph0), r1 = turn_frac(ph1);
                    const float s0 = __builtin_amdgcn_sinf(r0), c0 = __builtin_amdgcn_cosf(r0);
                    const float s1 = __builtin_amdgcn_sinf(r1), c1 = __builtin_amdgcn_cosf(r1);
                    split2(w0 * c0, w1 * c1, rh, rl);
                    split2(w0 * s0, w1 * s1, ih, il);
                }
                if (MIR && u == 0) { m_rh = rh; m_rl = rl; m_ih = ih; m_il = il; }
                if (u < 2) {
                    unsigned char* o = buf + sweep_row(u) * PK::ROWB0 + pp * 4 + 32 * hf;
                    *reinterpret_cast<uint32_t*>(o) = rh;
                    *reinterpret_cast<uint32_t*>(o + 2 * MF_KP) = ih;
                    *reinterpret_cast<uint32_t*>(o + PK::IMG) = rl;
                    *reinterpret_cast<uint32_t*>(o + PK::IMG + 2 * MF_KP) = il;
                } else {
                    unsigned char* o = buf + PK::IMG0 + grow * PK::ROWB1 + pp * 4 + 32 * hf;
                    *reinterpret_cast<uint32_t*>(o) = rh;
                    *reinterpret_cast<uint32_t*>(o + 16 * PK::ROWB1) = ih;
                    *reinterpret_cast<uint32_t*>(o + 32 * PK::ROWB1) = rh ^ 0x80008000u;
                    *reinterpret_cast<uint32_t*>(o + PK::IMG) = rl;
                    *reinterpret_cast<uint32_t*>(o + PK::IMG + 16 * PK::ROWB1) = il;
                    *reinterpret_cast<uint32_t*>(o + PK::IMG + 32 * PK::ROWB1) = rl ^ 0x80008000u;
                }
            }
        }
        fetch(next_panel);
    };

    // fragments of this wave's K step: 8 f16 of row (lane & 31), pixels 16 KS + 8 (lane >> 5) ..
    const int f0off = (lane & 31) * PK::ROWB0 + (lane >> 5) * 16 + 32 * KS;                      // first row tile, re plane
    const int f1off = PK::IMG0 + (lane & 31) * PK::ROWB1 + (lane >> 5) * 16 + 32 * KS;           // P: virtual row = lane & 31
    const int f2off = PK::IMG0 + (16 + (lane & 31)) * PK::ROWB1 + (lane >> 5) * 16 + 32 * KS;    // Q: im rows, then -re rows
    auto contract = [&](const unsigned char* buf) {
        auto ld = [&](int off) { return *reinterpret_cast<const uint4*>(buf + off); };
        uint4 sg = make_uint4(0, 0, 0, 0);
        if constexpr (SIGNED) sg = *reinterpret_cast<const uint4*>(buf + 2 * PK::IMG + (2 * KS + (lane >> 5)) * 16);
        auto sgn = [&](uint4 v) {
            if constexpr (SIGNED) { v.x ^= sg.x; v.y ^= sg.y; v.z ^= sg.z; v.w ^= sg.w; }
            return v;
        };
        if constexpr (!PAIR_B) {
            // tile (0,0), symmetric form:  Vr = A + A^T, A = (Lrh/2).Brh + (Lih/2).Bih + Lrh.Brl + Lih.Bil;
            //                              Vi = A - A^T, A = Lrh.Bih + Lrh.Bil - Lih.Brl   (transposes in the epilogue)
            // (the hi x hi products Lrh.Brh + Lih.Bih are symmetric by themselves: they go to an accumulator of their own,
            //  acc3, added once in the epilogue -- Vr = acc3 + acc0 + acc0^T -- instead of halving the L fragments in every K
            //  step: 8 v_pk_mul_f16 per K step less on the waves that also carry the extra generation sweep)
            const uint4 Brh = ld(f0off), Bih = ld(f0off + 2 * MF_KP), Brl = ld(f0off + PK::IMG), Bil = ld(f0off + PK::IMG + 2 * MF_KP);
            const uint4 Lrh = sgn(Brh), Lih = sgn(Bih);
            acc3 = RIME_MFMA(Lrh, Brh, acc3);
            acc1 = RIME_MFMA(Lrh, Bih, acc1);
            acc3 = RIME_MFMA(Lih, Bih, acc3);
            acc2 = RIME_MFMA(Lih, Brl, acc2);
            acc0 = RIME_MFMA(Lrh, Brl, acc0);
            acc1 = RIME_MFMA(Lrh, Bil, acc1);
            acc0 = RIME_MFMA(Lih, Bil, acc0);
        } else {
            const uint4 Lrh = sgn(ld(f0off)), Lih = sgn(ld(f0off + 2 * MF_KP));
            const uint4 Lrl = sgn(ld(f0off + PK::IMG)), Lil = sgn(ld(f0off + PK::IMG + 2 * MF_KP));
            const uint4 Ph = ld(f1off), Pl = ld(f1off + PK::IMG), Qh = ld(f2off), Ql = ld(f2off + PK::IMG);
            const uint4 SPh = sgn(Ph), SPl = sgn(Pl);
            acc0 = RIME_MFMA(Lrh, Ph, acc0);
            acc1 = RIME_MFMA(SPh, Ph, acc1);
            acc0 = RIME_MFMA(Lih, Qh, acc0);
            acc1 = RIME_MFMA(SPh, Pl, acc1);
            acc0 = RIME_MFMA(Lrh, Pl, acc0);
            acc1 = RIME_MFMA(SPl, Ph, acc1);
            acc0 = RIME_MFMA(Lih, Ql, acc0);
            acc0 = RIME_MFMA(Lrl, Ph, acc0);
            acc0 = RIME_MFMA(Lil, Qh, acc0);
        }
    };

    unsigned char* const buf0 = smem;
    unsigned char* const buf1 = smem + PK::BUF;
    fetch(pbeg);
    generate(buf0, min(pbeg + 1, pend - 1));
    __syncthreads();
    for (int panel = pbeg; panel < pend; panel += 2) {
        if (panel + 1 < pend) generate(buf1, min(panel + 2, pend - 1));
        contract(buf0);
        __syncthreads();
        if (panel + 1 < pend) {
            if (panel + 2 < pend) generate(buf0, min(panel + 3, pend - 1));
            contract(buf1);
        }
        __syncthreads();
    }

    // epilogue.  The waves of a pair hold the two K steps' partial sums of the same tiles: wave 0 finishes the real part
    // of tile (0,0), wave 1 its imaginary part, wave 2 tile (0,1), wave 3 tile (1,1); each hands the other unit's partial
    // tile to its partner through LDS ([e][lane] floats behind the four transposition tiles).
    float* dst = A.ws + (((size_t)split * A.Nt + t) * A.Nf + f) * 2 * A.Nbl;
    const float inv = 1.0f / scl;
    const int col = lane & 31;
    RIME_MFMA_SETTLE();
    float* tr = reinterpret_cast<float*>(smem) + W * (32 * 33);
    float* ex = reinterpret_cast<float*>(smem) + 4 * (32 * 33);
    f32x16 val, sym;                                 // sym: the part of the real unit that is symmetric as it stands
#pragma unroll
    for (int e = 0; e < 16; ++e) sym[e] = 0.f;
    if constexpr (!PAIR_B) {
        // give: the unit the partner finishes; keep: mine.  W = 0 keeps R (acc0, acc3), gives I = acc1 - acc2; W = 1 the
        // reverse (its two R parts go through its own exchange tile and a fifth one)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float r = acc0[e], i = acc1[e] - acc2[e];
            val[e] = W == 0 ? r : i;
            if constexpr (W == 0) { sym[e] = acc3[e]; ex[(0 * 16 + e) * 64 + lane] = i; }
            else { ex[(1 * 16 + e) * 64 + lane] = r; ex[(4 * 16 + e) * 64 + lane] = acc3[e]; }
        }
    } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            val[e] = W == 2 ? acc0[e] : acc1[e];
            ex[(W * 16 + e) * 64 + lane] = W == 2 ? acc1[e] : acc0[e];
        }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 16; ++e) val[e] += ex[((W ^ 1) * 16 + e) * 64 + lane];
    if constexpr (W == 0) {
#pragma unroll
        for (int e = 0; e < 16; ++e) sym[e] += ex[(4 * 16 + e) * 64 + lane];
    }

    if constexpr (!PAIR_B) {
        constexpr int im = W;                        // wave 0: real part, wave 1: imaginary part
#pragma unroll
        for (int e = 0; e < 16; ++e) tr[((e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)) * 33 + col] = val[e];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float tv = tr[col * 33 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)];
            val[e] = im ? val[e] - tv : (val[e] + tv) + sym[e];
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int i = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5), j = col;
            const float v = val[e] * inv;
            const int bd = A.pair_direct[i * MF_NA + j];
            if (bd >= 0) dst[(size_t)im * A.Nbl + bd] = v;
            const int bc = A.pair_conj[i * MF_NA + j];
            if (bc >= 0) dst[(size_t)im * A.Nbl + bc] = im ? -v : v;
        }
    } else if constexpr (W == 2) {
        // tile (0,1): rows = antennas 0..31, columns 0..15: Vr of antenna 32 + col, columns 16..31: Vi of antenna 16 + col
        const int im = col >> 4, j = 32 + (col & 15);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int i = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
            const float v = val[e] * inv;
            const int bd = A.pair_direct[i * MF_NA + j];
            if (bd >= 0) dst[(size_t)im * A.Nbl + bd] = v;
            const int bc = A.pair_conj[i * MF_NA + j];
            if (bc >= 0) dst[(size_t)im * A.Nbl + bc] = im ? -v : v;
        }
    } else {
        // tile (1,1): D rows a (elements e < 8) and 16 + a (e + 8) of this lane's column, partner column in lane ^ 16:
        //   col < 16:  Vr[a][col]      = D[a][col] + D[16 + a][16 + col]
        //   col >= 16: Vi[a][col - 16] = D[a][col] - D[16 + a][col - 16]
        const int im = col >> 4, j = 32 + (col & 15);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float other = __shfl_xor(val[e + 8], 16, 64);
            const float v = (im ? val[e] - other : val[e] + other) * inv;
            const int i = 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
            const int bd = A.pair_direct[i * MF_NA + j];
            if (bd >= 0) dst[(size_t)im * A.Nbl + bd] = v;
            const int bc = A.pair_conj[i * MF_NA + j];
            if (bc >= 0) dst[(size_t)im * A.Nbl + bc] = im ? -v : v;
        }
    }
}

template <bool SIGNED, bool MIR = false>
__global__ void __launch_bounds__(PK::NW * 64, 2)
fringe_ant_fwd_packed_kernel(AntArgs A)
{
    extern __shared__ __align__(16) unsigned char smem[];
    if (row_is_signed(A) != SIGNED) return;              // uniform over the block
    switch (threadIdx.x >> 6) {                          // wave-uniform: every wave runs the same barriers
        case 0: ant_fwd_packed_body<0, SIGNED, MIR>(A, smem); break;
        case 1: ant_fwd_packed_body<1, SIGNED, MIR>(A, smem); break;
        case 2: ant_fwd_packed_body<2, SIGNED, MIR>(A, smem); break;
        default: ant_fwd_packed_body<3, SIGNED, MIR>(A, smem); break;
    }
}

// ---------------------------------------------------------------------------------------
// CONJUGATE-PAIR FORM (round 5): the pair matrix of a point-symmetric array from the phasors of HALF its antennas
//
// MIRROR PAIRS above stop evaluating the conjugate phasors but still CONTRACT them: the mirror rows sit in the operand images and
// the matrix pipe multiplies the same numbers twice with a sign flipped.  With F "first" antennas (one of each pair; partner
// i' of first i has E_i' = conj(E_i)) and X = sqrt(|psky| scale) E of the firsts only, ALL pairs of the 2 F antennas are
//     A[i,j] = sum_p s_p conj(X_i) X_j = V[i , j]          V[i', j'] = conj(A[i,j])          (s_p = sign of psky)
//     B[i,j] = sum_p s_p      X_i  X_j = V[i', j]          V[i , j'] = conj(B[i,j]),   B[i,i] = V[i', i]
// and both come from the SAME three real products of the F-row images (Xr, Xi the planes):
//     Pcc = Xr s Xr^T,  Pss = Xi s Xi^T,  Pcs = Xr s Xi^T:      A = (Pcc + Pss) + i (Pcs - Pcs^T),   B = (Pcc - Pss) + i (Pcs + Pcs^T)
// i.e. the generic kernel's products of a 64-row block with the two halves of its real part kept in accumulators of their
// own: 26 MFMAs per K step for up to 128 antennas instead of 100 (12 on the off-diagonal tile, 7 on each diagonal tile),
// and 64 generated rows without the 64 conjugate copies.  Antennas without a partner take a row like a first (their B
// entries towards a mirror that does not exist have no baseline slot).  CEN: when the firsts and singles fill all 64 rows,
// ONE more antenna can still be served if it sits AT the centre of symmetry (the hub of a hexagon; the headline array is 63
// pairs + an outrigger + the hub): its phasor is 1, so its visibilities V[c, j] = sum_p psky_p E_j are column sums of
// the image, accumulated by the lanes that generate the rows (4 FMAs per generated pixel pair) and reduced in the epilogue.
// The result goes out through the pair tables of a VIRTUAL 128-row block -- rows 0..63 the firsts, row 64 + i the mirror
// of row i -- whose ten upper-triangular tiles are exactly A (tiles (0,0) (0,1) (1,1)), conj(A) ((2,2) (2,3) (3,3)) and
// conj(B) ((0,2) (0,3) (1,2) (1,3)); the host (ops._pair_block) builds them with the rule of every diagonal block.
// Deal: wave 0: Pcc, Pss of tile (0,1) (6 MFMAs per K step); wave 1: Pcs, Psc of tile (0,1) (6); wave 2: tile (0,0);
// wave 3: tile (1,1) (7 each: (hi / 2) x hi + hi x lo of Pcc and of Pss -- the other halves are their transposes --
// and the three products of Pcs in the two accumulators of the generic kernel's diagonal form).  164 registers: three
// blocks per CU.
// ---------------------------------------------------------------------------------------
struct PairArgs : AntArgs {
    const int* centre;         // CEN: [2][128] baseline slots receiving V[c, r] (first 128) / conj(V[c, r]) (last 128), r = virtual row
};

template <int W, bool SIGNED, bool CEN, bool FLAT>
__device__ __forceinline__ void pair_fwd_body(const PairArgs& A, unsigned char* smem)
{
    using SH = FwdShape<2, 2, false>;
    constexpr int MF_IMG = SH::IMG, MF_BUF = SH::BUF;
    const int tid = threadIdx.x, lane = tid & 63;
    const int f = __builtin_amdgcn_readfirstlane(blockIdx.x % A.Nf), ts = blockIdx.x / A.Nf;
    const int t = __builtin_amdgcn_readfirstlane(ts / A.S), split = __builtin_amdgcn_readfirstlane(ts % A.S);

    const double nu_c = A.sign * A.freqs[f] * (1.0 / 2.99792458e8);
    const float scl = A.scale[t * A.Nf + f];
    const float* arow = A.psky + (size_t)t * A.st_t + (size_t)f * A.st_f;
    const double* sd = A.sdir + (size_t)t * 3 * A.Pstride;
    const int st_p = __builtin_amdgcn_readfirstlane((int)A.st_p);

    // generation: the half-panel mapping of the two-tile blocks -- lane = (pixel pair pp, row ag of an octet), this wave writes
    // the 16-pixel half W & 1 for the octets of the wave pair W >> 1 (rows 32 (u >> 1) + 16 (W >> 1) + 8 (u & 1) + i)
    const int pp = lane & 7, ag = lane >> 3;
    constexpr int hf = W & 1;
    constexpr int NGEN = 4;
    const int orow = 16 * (W >> 1) + 2 * (ag & 3) + (ag >> 2);
    auto octet_row = [&](int u) { return 32 * (u >> 1) + 8 * (u & 1) + orow; };
    const int mrow = A.Nant - 16 * (W >> 1);
    const int nk = min(NGEN, (max(mrow, 0) + 31) / 32 + (max(mrow - 8, 0) + 31) / 32);     // sweeps whose octet starts below Nant
    double ax[NGEN], ay[NGEN], az[NGEN];
#pragma unroll
    for (int u = 0; u < NGEN; ++u) {
        const int an = octet_row(u);
        const bool ok = an < A.Nant;
        ax[u] = ok ? nu_c * A.antpos[3 * an] : 0.0;
        ay[u] = ok ? nu_c * A.antpos[3 * an + 1] : 0.0;
        az[u] = ok ? nu_c * A.antpos[3 * an + 2] : 0.0;
    }

    // waves 0, 1: acc[0], acc[1]; waves 2, 3: 0 half of Pcc, 1 half of Pss, 2 cs (hi x hi + hi x lo), 3 sc (hi x lo)
    constexpr int NACC = W < 2 ? 2 : 4;
    f32x16 acc[NACC];
#pragma unroll
    for (int s = 0; s < NACC; ++s)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[s][e] = 0.f;
    float cr[NGEN], ci[NGEN];                        // CEN: column sums of this lane's rows
#pragma unroll
    for (int u = 0; u < NGEN; ++u) { cr[u] = 0.f; ci[u] = 0.f; }

    const int npanel = A.Pstride / MF_KP;
    const int pbeg = __builtin_amdgcn_readfirstlane(split * A.panels_per_split);
    const int pend = __builtin_amdgcn_readfirstlane(min(npanel, pbeg + A.panels_per_split));
    if (pbeg >= pend) return;                        // uniform over the block

    // panel fetch through buffer loads (see the generic kernel): descriptor in SGPRs + constant lane offset + scalar offset
    double2 sx, sy, sz = make_double2(0.0, 0.0); float2 av;
    const uint32_t lo_s = 16u * pp, lo_a0 = 8u * pp * (uint32_t)st_p, lo_a1 = lo_a0 + 4u * (uint32_t)st_p;
    auto uniform_ptr = [](const void* q) {
        const unsigned long long a = reinterpret_cast<unsigned long long>(q);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
        return reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo);
    };
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        uniform_ptr(sd), 0, __builtin_amdgcn_readfirstlane((int)min((long long)3 * A.Pstride * 8, 0x7fffffffLL)), 0x00020000);
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        uniform_ptr(arow), 0, __builtin_amdgcn_readfirstlane((int)min((long long)A.Pstride * st_p * 4, 0x7fffffffLL)), 0x00020000);
    auto fetch = [&](int panel) {
        const int p0 = panel * MF_KP + 16 * hf;      // uniform
        sx = __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)lo_s, p0 * 8, 0));
        sy = __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)lo_s, (A.Pstride + p0) * 8, 0));
        if constexpr (!FLAT) sz = __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)lo_s, (2 * A.Pstride + p0) * 8, 0));
        const int so = p0 * st_p * 4;
        av = make_float2(__uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ra, (int)lo_a0, so, 0)),
                         __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ra, (int)lo_a1, so, 0)));
    };
    auto generate = [&](unsigned char* buf, int next_panel) {
        const float w0 = __builtin_amdgcn_sqrtf(fabsf(av.x) * scl), w1 = __builtin_amdgcn_sqrtf(fabsf(av.y) * scl);
        if (SIGNED && W < 2 && lane < 8)
            *reinterpret_cast<uint32_t*>(buf + 2 * MF_IMG + 4 * (8 * hf + pp)) =
                ((__float_as_uint(av.x) >> 16) & 0x8000u) | (__float_as_uint(av.y) & 0x80000000u);
        // CEN: sum_p psky scale E = sum_p (+-w)(w E): the weight once more, with the sign of psky
        float g0 = w0, g1 = w1;
        if constexpr (CEN && SIGNED) {
            g0 = __uint_as_float(__float_as_uint(w0) | (__float_as_uint(av.x) & 0x80000000u));
            g1 = __uint_as_float(__float_as_uint(w1) | (__float_as_uint(av.y) & 0x80000000u));
        }
#pragma unroll
        for (int u = 0; u < NGEN; ++u) {
            if (u < nk) {
                const double ph0 = phase_of<FLAT>(ax[u], sx.x, ay[u], sy.x, az[u], sz.x);
                const double ph1 = phase_of<FLAT>(ax[u], sx.y, ay[u], sy.y, az[u], sz.y);
                const float r0 = turn_frac(ph0), r1 = turn_frac(ph1);
                const float s0 = __builtin_amdgcn_sinf(r0), c0 = __builtin_amdgcn_cosf(r0);
                const float s1 = __builtin_amdgcn_sinf(r1), c1 = __builtin_amdgcn_cosf(r1);
                uint32_t rh, rl, ih, il;
                if constexpr (!CEN) {
                    split2(w0 * c0, w1 * c1, rh, rl);
                    split2(w0 * s0, w1 * s1, ih, il);
                } else {
                    float xr0, xr1, xi0, xi1;
                    split2_prod(w0, c0, w1, c1, xr0, xr1, rh, rl);
                    split2_prod(w0, s0, w1, s1, xi0, xi1, ih, il);
                    cr[u] = fmaf(g0, xr0, cr[u]); keep_scalar(cr[u]); ci[u] = fmaf(g0, xi0, ci[u]);
                    cr[u] = fmaf(g1, xr1, cr[u]); keep_scalar(cr[u]); ci[u] = fmaf(g1, xi1, ci[u]);
                }
                unsigned char* o = buf + octet_row(u) * MF_ROWB + pp * 4 + 32 * hf;
                *reinterpret_cast<uint32_t*>(o) = rh;
                *reinterpret_cast<uint32_t*>(o + 2 * MF_KP) = ih;
                *reinterpret_cast<uint32_t*>(o + MF_IMG) = rl;
                *reinterpret_cast<uint32_t*>(o + MF_IMG + 2 * MF_KP) = il;
            }
        }
        fetch(next_panel);
    };

    const int foff = (lane & 31) * MF_ROWB + (lane >> 5) * 16;   // fragment: row, k-half
    auto contract = [&](const unsigned char* buf) {
        auto frag = [&](int tile, int img, int im, int ks) {
            return *reinterpret_cast<const uint4*>(buf + img * MF_IMG + tile * 32 * MF_ROWB + foff + im * 2 * MF_KP + 32 * ks);
        };
        auto sgn = [&](uint4 v, const uint4& sg) {
            if constexpr (SIGNED) { v.x ^= sg.x; v.y ^= sg.y; v.z ^= sg.z; v.w ^= sg.w; }
            return v;
        };
#pragma unroll
        for (int ks = 0; ks < MF_NH; ++ks) {
            uint4 sg = make_uint4(0, 0, 0, 0);
            if constexpr (SIGNED) sg = *reinterpret_cast<const uint4*>(buf + 2 * MF_IMG + (2 * ks + (lane >> 5)) * 16);
            if constexpr (W == 0) {                  // tile (0,1): Pcc -> acc[0], Pss -> acc[1]
                const uint4 Lrh = sgn(frag(0, 0, 0, ks), sg), Lih = sgn(frag(0, 0, 1, ks), sg);
                const uint4 Lrl = sgn(frag(0, 1, 0, ks), sg), Lil = sgn(frag(0, 1, 1, ks), sg);
                const uint4 Brh = frag(1, 0, 0, ks), Bih = frag(1, 0, 1, ks), Brl = frag(1, 1, 0, ks), Bil = frag(1, 1, 1, ks);
                acc[0] = RIME_MFMA(Lrh, Brh, acc[0]);
                acc[1] = RIME_MFMA(Lih, Bih, acc[1]);
                acc[0] = RIME_MFMA(Lrh, Brl, acc[0]);
                acc[1] = RIME_MFMA(Lih, Bil, acc[1]);
                acc[0] = RIME_MFMA(Lrl, Brh, acc[0]);
                acc[1] = RIME_MFMA(Lil, Bih, acc[1]);
            } else if constexpr (W == 1) {           // tile (0,1): Pcs = Lr.Bi -> acc[0], Psc = Li.Br -> acc[1]
                const uint4 Lrh = sgn(frag(0, 0, 0, ks), sg), Lih = sgn(frag(0, 0, 1, ks), sg);
                const uint4 Lrl = sgn(frag(0, 1, 0, ks), sg), Lil = sgn(frag(0, 1, 1, ks), sg);
                const uint4 Brh = frag(1, 0, 0, ks), Bih = frag(1, 0, 1, ks), Brl = frag(1, 1, 0, ks), Bil = frag(1, 1, 1, ks);
                acc[0] = RIME_MFMA(Lrh, Bih, acc[0]);
                acc[1] = RIME_MFMA(Lih, Brh, acc[1]);
                acc[0] = RIME_MFMA(Lrh, Bil, acc[0]);
                acc[1] = RIME_MFMA(Lih, Brl, acc[1]);
                acc[0] = RIME_MFMA(Lrl, Bih, acc[0]);
                acc[1] = RIME_MFMA(Lil, Brh, acc[1]);
            } else {                                 // diagonal tile W - 2: L and B are the same rows (up to the pixel sign)
                constexpr int tt = W - 2;
                const uint4 Brh = frag(tt, 0, 0, ks), Bih = frag(tt, 0, 1, ks), Brl = frag(tt, 1, 0, ks), Bil = frag(tt, 1, 1, ks);
                const uint4 Lrh = sgn(Brh, sg), Lih = sgn(Bih, sg);
                // Pcc = h + h^T with h = (Lrh / 2).Brh + Lrh.Brl (the lo x hi product is the transpose of hi x lo), Pss alike: the
                // halved fragments cost 8 v_pk_mul_f16 per K step and save two accumulators -- 164 registers instead of 196, i.e.
                // THREE blocks per CU instead of two, which is worth 6 % here (profiles/r05/pair_form.txt)
                const uint4 Hr = half_frag(Lrh), Hi = half_frag(Lih);
                acc[0] = RIME_MFMA(Hr, Brh, acc[0]);
                acc[1] = RIME_MFMA(Hi, Bih, acc[1]);
                acc[2] = RIME_MFMA(Lrh, Bih, acc[2]);
                acc[0] = RIME_MFMA(Lrh, Brl, acc[0]);
                acc[1] = RIME_MFMA(Lih, Bil, acc[1]);
                acc[3] = RIME_MFMA(Lih, Brl, acc[3]);            // its transpose = Lrl.Bih
                acc[2] = RIME_MFMA(Lrh, Bil, acc[2]);
            }
        }
    };

    unsigned char* const buf0 = smem;
    unsigned char* const buf1 = smem + MF_BUF;
    fetch(pbeg);
    generate(buf0, min(pbeg + 1, pend - 1));
    __syncthreads();
    for (int panel = pbeg; panel < pend; panel += 2) {
        if (panel + 1 < pend) generate(buf1, min(panel + 2, pend - 1));
        contract(buf0);
        __syncthreads();
        if (panel + 1 < pend) {
            if (panel + 2 < pend) generate(buf0, min(panel + 3, pend - 1));
            contract(buf1);
        }
        __syncthreads();
    }

    // epilogue: slab ws[split][t][f][re|im][Nbl] as in the generic kernel, through the tables of the virtual 128-row block
    float* dst = A.ws + (((size_t)split * A.Nt + t) * A.Nf + f) * 2 * A.Nbl;
    const float inv = 1.0f / scl;
    const int col = lane & 31, rb = 4 * (lane >> 5);
    RIME_MFMA_SETTLE();
    auto put = [&](int r, int c, float vr, float vi) {            // V[r, c] = vr + i vi
        const int bd = A.pair_direct[r * MF_NA + c];
        if (bd >= 0) { dst[bd] = vr; dst[(size_t)A.Nbl + bd] = vi; }
        const int bc = A.pair_conj[r * MF_NA + c];
        if (bc >= 0) { dst[bc] = vr; dst[(size_t)A.Nbl + bc] = -vi; }
    };
    auto put1 = [&](int r, int c, int im, float v) {              // one plane of V[r, c]
        const int bd = A.pair_direct[r * MF_NA + c];
        if (bd >= 0) dst[(size_t)im * A.Nbl + bd] = v;
        const int bc = A.pair_conj[r * MF_NA + c];
        if (bc >= 0) dst[(size_t)im * A.Nbl + bc] = im ? -v : v;
    };
    if constexpr (CEN) {
        // the hub's visibilities: sum the column sums over the 8 pixel-pair lanes and the two half-panel waves of a row
#pragma unroll
        for (int u = 0; u < NGEN; ++u)
#pragma unroll
            for (int o = 1; o < 8; o <<= 1) { cr[u] += __shfl_xor(cr[u], o, 64); keep_scalar(cr[u]); ci[u] += __shfl_xor(ci[u], o, 64); keep_scalar(ci[u]); }
        float* cs = reinterpret_cast<float*>(smem) + 4 * (32 * 33);          // [half][64 rows][re | im], behind the transposition tiles
        if (pp == 0) {
#pragma unroll
            for (int u = 0; u < NGEN; ++u)
                if (u < nk) { cs[(hf * 64 + octet_row(u)) * 2] = cr[u]; cs[(hf * 64 + octet_row(u)) * 2 + 1] = ci[u]; }
        }
        __syncthreads();
        if (tid < A.Nant) {                          // (Nant <= 64: wave 0)
            float vr = cs[tid * 2] + cs[(64 + tid) * 2], vi = cs[tid * 2 + 1] + cs[(64 + tid) * 2 + 1];
            keep_scalar(vr); keep_scalar(vi);
            vr *= inv; keep_scalar(vr); vi *= inv;
            // V[c, j] = sum psky E_j for the first in row j, its conjugate for the mirror in row 64 + j
            int b = A.centre[tid];
            if (b >= 0) { dst[b] = vr; dst[(size_t)A.Nbl + b] = vi; }
            b = A.centre[MF_NA + tid];
            if (b >= 0) { dst[b] = vr; dst[(size_t)A.Nbl + b] = -vi; }
            b = A.centre[64 + tid];
            if (b >= 0) { dst[b] = vr; dst[(size_t)A.Nbl + b] = -vi; }
            b = A.centre[MF_NA + 64 + tid];
            if (b >= 0) { dst[b] = vr; dst[(size_t)A.Nbl + b] = vi; }
        }
    }
    // a private 32 x 33 float tile per wave takes the transposes (the image buffers are free after the loop's last barrier)
    float* tr = reinterpret_cast<float*>(smem) + W * (32 * 33);
    auto transposed = [&](const f32x16& v) {
        f32x16 tv;
#pragma unroll
        for (int e = 0; e < 16; ++e) tr[((e & 3) + 8 * (e >> 2) + rb) * 33 + col] = v[e];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int e = 0; e < 16; ++e) tv[e] = tr[col * 33 + (e & 3) + 8 * (e >> 2) + rb];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        return tv;
    };
    // Store order.  A wave's store instruction covers one accumulator element of all lanes: 32 consecutive COLUMNS of one row.
    // With the baselines in antenna order (the usual sim_bls) the slots of V[r, c .. c + 31] are consecutive when the lanes run
    // along the LATER antenna of the pair: true as they stand for V[i, j] and V[i, j'] (rows i, lanes j), but V[i', j'] and
    // V[j, i'] of the off-diagonal tile want the lanes along i -- those two go out from the TRANSPOSED tile (3.5 -> 2.3 GB of slab
    // traffic per headline launch); on a diagonal tile B = B^T lets element (r, c) write V[r, c'] itself.
    if constexpr (W < 2) {
        // tile (0,1), element (i, j): A[i,j] -> V[i, j] and conj -> V[i', j'];  conj(B[i,j]) -> V[j, i'] and V[i, j']
        f32x16 s, d;
#pragma unroll
        for (int e = 0; e < 16; ++e) { s[e] = (acc[0][e] + acc[1][e]) * inv; d[e] = (acc[0][e] - acc[1][e]) * inv; }
        const f32x16 st = transposed(s), dt = transposed(d);          // st[e] of lane c = s at (row c, column row(e))
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int re = (e & 3) + 8 * (e >> 2) + rb;
            const int i = re, j = 32 + col;                               // as accumulated: rows i, lanes j
            const int it = col, jt = 32 + re;                             // transposed: rows j, lanes i
            if constexpr (W == 0) {                  // real parts: Ar = Pcc + Pss, Br = Pcc - Pss
                put1(i, j, 0, s[e]); put1(i, 64 + j, 0, d[e]);
                put1(64 + it, 64 + jt, 0, st[e]); put1(jt, 64 + it, 0, dt[e]);
            } else {                                 // imaginary parts: Ai = Pcs - Psc (d), Bi = Pcs + Psc (s)
                put1(i, j, 1, d[e]); put1(i, 64 + j, 1, -s[e]);
                put1(64 + it, 64 + jt, 1, -dt[e]); put1(jt, 64 + it, 1, -st[e]);
            }
        }
    } else {
        constexpr int tt = W - 2;
        // Pcc = h + h^T (acc 0), Pss alike (acc 1);  Pcs = acc2 + acc3^T,  Pcs^T = acc2^T + acc3
        f32x16 ar, br, ai, bi, x, y;
#pragma unroll
        for (int e = 0; e < 16; ++e) { x[e] = acc[0][e] + acc[1][e]; y[e] = acc[0][e] - acc[1][e]; }
        const f32x16 xt = transposed(x), yt = transposed(y);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            ar[e] = x[e] + xt[e];
            br[e] = y[e] + yt[e];
            x[e] = acc[2][e] - acc[3][e]; y[e] = acc[2][e] + acc[3][e];
        }
        const f32x16 xt2 = transposed(x), yt2 = transposed(y);
#pragma unroll
        for (int e = 0; e < 16; ++e) { ai[e] = x[e] - xt2[e]; bi[e] = y[e] + yt2[e]; }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int i = 32 * tt + (e & 3) + 8 * (e >> 2) + rb, j = 32 * tt + col;
            put(i, j, ar[e] * inv, ai[e] * inv);
            put(64 + i, 64 + j, ar[e] * inv, -ai[e] * inv);
            put(i, 64 + j, br[e] * inv, -bi[e] * inv);            // conj(B[j,i]) = conj(B[i,j]) -> V[i, j']
        }
    }
}

template <bool SIGNED, bool CEN, bool FLAT>
__global__ void __launch_bounds__(256, 3)
fringe_pair_fwd_kernel(PairArgs A)
{
    extern __shared__ __align__(16) unsigned char smem[];
    if (row_is_signed(A) != SIGNED) return;              // uniform over the block
    switch (threadIdx.x >> 6) {                          // wave-uniform: every wave runs the same barriers
        case 0: pair_fwd_body<0, SIGNED, CEN, FLAT>(A, smem); break;
        case 1: pair_fwd_body<1, SIGNED, CEN, FLAT>(A, smem); break;
        case 2: pair_fwd_body<2, SIGNED, CEN, FLAT>(A, smem); break;
        default: pair_fwd_body<3, SIGNED, CEN, FLAT>(A, smem); break;
    }
}

// Conjugate-pair form, ONE row tile (<= 32 rows; arrays of 33..64 antennas -- HERA-37: 18 pairs + the hub in 19 rows): the single
// diagonal tile costs 7 MFMAs per K step (the packed 33..48-antenna kernel: 16, the generic two-tile kernel: 26).  Deal as for
// the generic one-tile shape: a wave takes ONE K step of every panel (W & 1); waves 0, 1 the halves of Pcc and Pss (4 MFMAs),
// waves 2, 3 the two accumulators of Pcs (3); the odd wave of a pair hands its partial tiles to the even one in the epilogue.
// No hub path: an array whose rows exceed 32 takes the two-tile kernel.
template <int W, bool SIGNED, bool FLAT>
__device__ __forceinline__ void pair_fwd1_body(const PairArgs& A, unsigned char* smem)
{
    using SH = FwdShape<1, 1, false>;
    constexpr int MF_IMG = SH::IMG, MF_BUF = SH::BUF;
    constexpr int KS = W & 1;
    const int tid = threadIdx.x, lane = tid & 63;
    const int f = __builtin_amdgcn_readfirstlane(blockIdx.x % A.Nf), ts = blockIdx.x / A.Nf;
    const int t = __builtin_amdgcn_readfirstlane(ts / A.S), split = __builtin_amdgcn_readfirstlane(ts % A.S);

    const double nu_c = A.sign * A.freqs[f] * (1.0 / 2.99792458e8);
    const float scl = A.scale[t * A.Nf + f];
    const float* arow = A.psky + (size_t)t * A.st_t + (size_t)f * A.st_f;
    const double* sd = A.sdir + (size_t)t * 3 * A.Pstride;
    const int st_p = __builtin_amdgcn_readfirstlane((int)A.st_p);

    // generation: this wave writes the 16-pixel half W & 1 for the two octets of the wave pair W >> 1 (rows 16 (W >> 1) + 8 u + i)
    const int pp = lane & 7, ag = lane >> 3;
    constexpr int hf = W & 1;
    constexpr int NGEN = 2;
    const int orow = 16 * (W >> 1) + 2 * (ag & 3) + (ag >> 2);
    const int nk = (16 * (W >> 1) < A.Nant) + (16 * (W >> 1) + 8 < A.Nant);          // sweeps whose octet starts below Nant
    double ax[NGEN], ay[NGEN], az[NGEN];
#pragma unroll
    for (int u = 0; u < NGEN; ++u) {
        const int an = 8 * u + orow;
        const bool ok = an < A.Nant;
        ax[u] = ok ? nu_c * A.antpos[3 * an] : 0.0;
        ay[u] = ok ? nu_c * A.antpos[3 * an + 1] : 0.0;
        az[u] = ok ? nu_c * A.antpos[3 * an + 2] : 0.0;
    }
    f32x16 acc0, acc1;                               // waves 0, 1: halves of Pcc, Pss; waves 2, 3: cs (hi x hi + hi x lo), sc (hi x lo)
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }

    const int npanel = A.Pstride / MF_KP;
    const int pbeg = __builtin_amdgcn_readfirstlane(split * A.panels_per_split);
    const int pend = __builtin_amdgcn_readfirstlane(min(npanel, pbeg + A.panels_per_split));
    if (pbeg >= pend) return;                        // uniform over the block

    double2 sx, sy, sz = make_double2(0.0, 0.0); float2 av;
    const uint32_t lo_s = 16u * pp, lo_a0 = 8u * pp * (uint32_t)st_p, lo_a1 = lo_a0 + 4u * (uint32_t)st_p;
    auto uniform_ptr = [](const void* q) {
        const unsigned long long a = reinterpret_cast<unsigned long long>(q);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
        return reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo);
    };
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        uniform_ptr(sd), 0, __builtin_amdgcn_readfirstlane((int)min((long long)3 * A.Pstride * 8, 0x7fffffffLL)), 0x00020000);
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        uniform_ptr(arow), 0, __builtin_amdgcn_readfirstlane((int)min((long long)A.Pstride * st_p * 4, 0x7fffffffLL)), 0x00020000);
    auto fetch = [&](int panel) {
        const int p0 = panel * MF_KP + 16 * hf;      // uniform
        sx = __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)lo_s, p0 * 8, 0));
        sy = __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)lo_s, (A.Pstride + p0) * 8, 0));
        if constexpr (!FLAT) sz = __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)lo_s, (2 * A.Pstride + p0) * 8, 0));
        const int so = p0 * st_p * 4;
        av = make_float2(__uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ra, (int)lo_a0, so, 0)),
                         __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ra, (int)lo_a1, so, 0)));
    };
    auto generate = [&](unsigned char* buf, int next_panel) {
        const float w0 = __builtin_amdgcn_sqrtf(fabsf(av.x) * scl), w1 = __builtin_amdgcn_sqrtf(fabsf(av.y) * scl);
        if (SIGNED && W < 2 && lane < 8)
            *reinterpret_cast<uint32_t*>(buf + 2 * MF_IMG + 4 * (8 * hf + pp)) =
                ((__float_as_uint(av.x) >> 16) & 0x8000u) | (__float_as_uint(av.y) & 0x80000000u);
#pragma unroll
        for (int u = 0; u < NGEN; ++u) {
            if (u < nk) {
                const double ph0 = phase_of<FLAT>(ax[u], sx.x, ay[u], sy.x, az[u], sz.x);
                const double ph1 = phase_of<FLAT>(ax[u], sx.y, ay[u], sy.y, az[u], sz.y);
                const float r0 = turn_frac(ph0), r1 = turn_frac(ph1);
                const float s0 = __builtin_amdgcn_sinf(r0), c0 = __builtin_amdgcn_cosf(r0);
                const float s1 = __builtin_amdgcn_sinf(r1), c1 = __builtin_amdgcn_cosf(r1);
                uint32_t rh, rl, ih, il;
                split2(w0 * c0, w1 * c1, rh, rl);
                split2(w0 * s0, w1 * s1, ih, il);
                unsigned char* o = buf + (8 * u + orow) * MF_ROWB + pp * 4 + 32 * hf;
                *reinterpret_cast<uint32_t*>(o) = rh;
                *reinterpret_cast<uint32_t*>(o + 2 * MF_KP) = ih;
                *reinterpret_cast<uint32_t*>(o + MF_IMG) = rl;
                *reinterpret_cast<uint32_t*>(o + MF_IMG + 2 * MF_KP) = il;
            }
        }
        fetch(next_panel);
    };

    const int foff = (lane & 31) * MF_ROWB + (lane >> 5) * 16 + 32 * KS;      // fragment of this wave's K step: row, k-half
    auto contract = [&](const unsigned char* buf) {
        auto frag = [&](int img, int im) { return *reinterpret_cast<const uint4*>(buf + img * MF_IMG + foff + im * 2 * MF_KP); };
        uint4 sg = make_uint4(0, 0, 0, 0);
        if constexpr (SIGNED) sg = *reinterpret_cast<const uint4*>(buf + 2 * MF_IMG + (2 * KS + (lane >> 5)) * 16);
        auto sgn = [&](uint4 v) {
            if constexpr (SIGNED) { v.x ^= sg.x; v.y ^= sg.y; v.z ^= sg.z; v.w ^= sg.w; }
            return v;
        };
        if constexpr (W < 2) {
            const uint4 Brh = frag(0, 0), Bih = frag(0, 1), Brl = frag(1, 0), Bil = frag(1, 1);
            const uint4 Lrh = sgn(Brh), Lih = sgn(Bih);
            const uint4 Hr = half_frag(Lrh), Hi = half_frag(Lih);
            acc0 = RIME_MFMA(Hr, Brh, acc0);
            acc1 = RIME_MFMA(Hi, Bih, acc1);
            acc0 = RIME_MFMA(Lrh, Brl, acc0);
            acc1 = RIME_MFMA(Lih, Bil, acc1);
        } else {
            const uint4 Brh = frag(0, 0), Bih = frag(0, 1), Brl = frag(1, 0), Bil = frag(1, 1);
            const uint4 Lrh = sgn(Brh), Lih = sgn(Bih);
            acc0 = RIME_MFMA(Lrh, Bih, acc0);
            acc1 = RIME_MFMA(Lih, Brl, acc1);
            acc0 = RIME_MFMA(Lrh, Bil, acc0);
        }
    };

    unsigned char* const buf0 = smem;
    unsigned char* const buf1 = smem + MF_BUF;
    fetch(pbeg);
    generate(buf0, min(pbeg + 1, pend - 1));
    __syncthreads();
    for (int panel = pbeg; panel < pend; panel += 2) {
        if (panel + 1 < pend) generate(buf1, min(panel + 2, pend - 1));
        contract(buf0);
        __syncthreads();
        if (panel + 1 < pend) {
            if (panel + 2 < pend) generate(buf0, min(panel + 3, pend - 1));
            contract(buf1);
        }
        __syncthreads();
    }

    // epilogue: the odd wave of a pair hands its K step's partial tiles to the even one ([e][lane] floats behind the transposition
    // tiles); wave 0 finishes the real parts, wave 2 the imaginary parts
    float* dst = A.ws + (((size_t)split * A.Nt + t) * A.Nf + f) * 2 * A.Nbl;
    const float inv = 1.0f / scl;
    const int col = lane & 31, rb = 4 * (lane >> 5);
    RIME_MFMA_SETTLE();
    float* ex = reinterpret_cast<float*>(smem) + 4 * (32 * 33) + (W >> 1) * (2 * 16 * 64);
    if constexpr ((W & 1) == 1) {
#pragma unroll
        for (int e = 0; e < 16; ++e) { ex[e * 64 + lane] = acc0[e]; ex[(16 + e) * 64 + lane] = acc1[e]; }
    }
    __syncthreads();
    if constexpr ((W & 1) == 1) return;
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc0[e] += ex[e * 64 + lane]; acc1[e] += ex[(16 + e) * 64 + lane]; }
    float* tr = reinterpret_cast<float*>(smem) + W * (32 * 33);
    auto transposed = [&](const f32x16& v) {
        f32x16 tv;
#pragma unroll
        for (int e = 0; e < 16; ++e) tr[((e & 3) + 8 * (e >> 2) + rb) * 33 + col] = v[e];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int e = 0; e < 16; ++e) tv[e] = tr[col * 33 + (e & 3) + 8 * (e >> 2) + rb];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        return tv;
    };
    auto put1 = [&](int r, int c, int im, float v) {              // one plane of V[r, c] of the virtual 128-row block
        const int bd = A.pair_direct[r * MF_NA + c];
        if (bd >= 0) dst[(size_t)im * A.Nbl + bd] = v;
        const int bc = A.pair_conj[r * MF_NA + c];
        if (bc >= 0) dst[(size_t)im * A.Nbl + bc] = im ? -v : v;
    };
    f32x16 x, y;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        if constexpr (W == 0) { x[e] = acc0[e] + acc1[e]; y[e] = acc0[e] - acc1[e]; }     // halves of Pcc + Pss, Pcc - Pss
        else { x[e] = acc0[e] - acc1[e]; y[e] = acc0[e] + acc1[e]; }                      // Pcs -+ Pcs^T before the transposes
    }
    const f32x16 xt = transposed(x), yt = transposed(y);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int i = (e & 3) + 8 * (e >> 2) + rb, j = col;
        if constexpr (W == 0) {
            const float ar = (x[e] + xt[e]) * inv, br = (y[e] + yt[e]) * inv;
            put1(i, j, 0, ar); put1(64 + i, 64 + j, 0, ar); put1(i, 64 + j, 0, br);     // (B = B^T: lanes along the later antenna)
        } else {
            const float ai = (x[e] - xt[e]) * inv, bi = (y[e] + yt[e]) * inv;
            put1(i, j, 1, ai); put1(64 + i, 64 + j, 1, -ai); put1(i, 64 + j, 1, -bi);
        }
    }
}

template <bool SIGNED, bool FLAT>
__global__ void __launch_bounds__(256, 2)
fringe_pair_fwd1_kernel(PairArgs A)
{
    extern __shared__ __align__(16) unsigned char smem[];
    if (row_is_signed(A) != SIGNED) return;              // uniform over the block
    switch (threadIdx.x >> 6) {                          // wave-uniform: every wave runs the same barriers
        case 0: pair_fwd1_body<0, SIGNED, FLAT>(A, smem); break;
        case 1: pair_fwd1_body<1, SIGNED, FLAT>(A, smem); break;
        case 2: pair_fwd1_body<2, SIGNED, FLAT>(A, smem); break;
        default: pair_fwd1_body<3, SIGNED, FLAT>(A, smem); break;
    }
}

// ---------------------------------------------------------------------------------------
// backward:  gpsky[t,f,p] = Re sum_{i,j} E_i(p) conj(E_j(p)) G[i,j]
//                        = sum_i ( Er_i Tr_i + Ei_i Ti_i ),   T_i(p) = sum_j conj(G[i,j]) E_j(p)
// G[i,j] (tile(i) <= tile(j)) collects gvis of pair (i -> j) and the conjugate of pair (j -> i)
// through the same two tables as the forward.  T is a block-upper-triangular complex GEMM
// (M = antennas i, N = pixels, K = antennas j) on v_mfma_f32_32x32x16_f16 with the same hi/lo
// f16 split.  Planar K layout (16 antennas per MFMA):
//     Tr += Gr.Er + Gi.Ei,     Ti += Gr.Ei + (-Gi).Er
// G (scaled by a power of two per (t,f)) sits in LDS for the whole block as six planes in
// A-fragment order (Gr, Gi, -Gi; hi and lo), so no operand is rotated or negated in the loop (VALU
// and MFMA issue do not overlap on gfx950, see the forward kernel).  E fragments are generated in
// registers by the lane that consumes them (lane = pixel, so every E value is computed exactly
// once); the K index of a fragment is mapped to antennas so that the D rows a lane holds are
// exactly the antennas it generated, which makes the final contraction with E_i lane-local.
// No atomics.  History at the C4 shape: interleaved (re,im) K layout with rot90 of the G fragments
// on the fly: 11.1 ms; this kernel 10.3 ms.  Tried and dropped: the tile count as a template
// parameter (the accumulator zero-fill folds into the first MFMAs, 103 -> 22 v_mov, but the
// schedule spills 9 registers: 3 % slower; with a sched_barrier between row tiles it does not spill and
// runs 13 % fewer VALU instructions -- at the same speed: moves and integer adds ride in the MFMA
// shadow, the f64 phase and sin/cos instructions are what the matrix pipe waits for.  But replacing
// v_cvt_f32_f64 of the fraction by v_alignbit + add + compare + select is 5 % SLOWER in both kernels).
// ---------------------------------------------------------------------------------------
struct AntBwdArgs {
    const double* antpos; const double* sdir; const double* freqs;
    const float* gvis;         // [Nbl, Nt, Nf, 2]
    const float* gvt;          // workspace: gvis transposed to [Nt, Nf, re|im, Nbl] (coalesced staging)
    const float* gscale;       // [Nt, Nf] power-of-two pre-scale of gvis
    const int* pair_direct; const int* pair_conj;
    float* gpsky;              // strided [t][f][p]
    int Nant, Nbl, Nt, Nf, Pstride;
    int S, tiles_per_split;    // pixel tiles (32 px) per block
    int accumulate;            // add to gpsky instead of overwriting (blocks after the first)
    int rows_i;                // cross blocks: antpos rows of group I (group J follows)
    long long st_t, st_f, st_p;
    double sign;
    float imsign;              // complex psky: sign of the imaginary-plane gradient (-1: block contracted conj(psky))
    int mirror;                // as in AntArgs: bit g = (row tile, K step) whose second octet is the conjugate of its first
};

constexpr int MB_TILES = 10;                        // upper-triangular 32x32 tiles of a 128x128 matrix
constexpr int MB_PLANE = MB_TILES * 2 * 2 * 32 * 16;  // bytes per G plane: (tile, ks, h, row) x 8 f16 = 20480
constexpr size_t MB_LDS = 6 * (size_t)MB_PLANE + MF_NA * 3 * sizeof(double);

__device__ __forceinline__ int tri_index(int ti, int tj) { return ti * 4 - ti * (ti - 1) / 2 + (tj - ti); }

// Staging of a diagonal block of the backward: antenna coordinates (x sign nu / c) and the six G planes in A-fragment order into
// LDS (NT = threads of the block).
template <bool CPLX, int NT>
__device__ __forceinline__ void bwd_stage_diag(const AntBwdArgs& A, unsigned char* g_img, double* ant_lds, int t, int f, int TA, float& gs)
{
    const int tid = threadIdx.x;
    const double nu_c = A.sign * A.freqs[f] * (1.0 / 2.99792458e8);
    for (int i = tid; i < MF_NA * 3; i += NT)
        ant_lds[i] = (i < A.Nant * 3) ? nu_c * A.antpos[i] : 0.0;

    // stage G: fragment element (tile, ks, h, row, jj) <-> pair (i = 32 ti + row,
    // j = 32 tj + (jj & 3) + 8 (2 ks + (jj >> 2)) + 4 h); a thread packs the (jj, jj + 1) pair
    gs = A.gscale[t * A.Nf + f];
    // tiles (ti, tj) with ti <= tj < TA are the ones the contraction reads: in the 4 x 4 upper-triangular numbering they end at
    // index 0 / 4 / 7 / 9 for TA = 1 .. 4, so small arrays stage 1 / 5 / 8 tiles instead of 10
    const int ntl = TA == 1 ? 1 : TA == 2 ? 5 : TA == 3 ? 8 : MB_TILES;
    for (int e = tid; e < ntl * 2 * 2 * 32 * 4; e += NT) {
        const int jp = e & 3, row = (e >> 2) & 31, h = (e >> 7) & 1, ks = (e >> 8) & 1, tile = e >> 9;
        int ti = 0, rem = tile;
        while (rem >= 4 - ti) { rem -= 4 - ti; ++ti; }
        const int tj = ti + rem;
        const int i = 32 * ti + row;
        const int j0 = 32 * tj + ((2 * jp) & 3) + 8 * (2 * ks + (jp >> 1)) + 4 * h;
        float gr[2] = {0.f, 0.f}, gi[2] = {0.f, 0.f};
        if (ti < TA && tj < TA) {
            const float* gre = A.gvt + ((size_t)t * A.Nf + f) * 2 * A.Nbl;
            const float* gim = gre + A.Nbl;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int bd = A.pair_direct[i * MF_NA + j0 + q];
                if (bd >= 0) { gr[q] += gre[bd]; gi[q] += gim[bd]; }
                const int bc = A.pair_conj[i * MF_NA + j0 + q];
                if (bc >= 0) { gr[q] += gre[bc]; gi[q] -= gim[bc]; }
            }
            if constexpr (!CPLX) {
                if (ti == tj) {
                    // diagonal tile, real psky: Re(E^H conj(G) E) = 1/2 E^H H E with H = conj(G) + conj(G)^H Hermitian
                    //   = Er^T S Er + Ei^T S Ei + Ei^T A Er,   S = (Gr + Gr^T) / 2 (symmetric),  A = Gi^T - Gi (antisymmetric)
                    // i.e. accR = S Er, accI = S Ei + A Er: three real products instead of four (9 MFMAs, not 12).  The
                    // planes hold S in place of Gr and A in place of -Gi; the Gi planes of these tiles are not read.
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        float tr = 0.f, ti_ = 0.f;               // G[j, i]
                        const int bd = A.pair_direct[(j0 + q) * MF_NA + i];
                        if (bd >= 0) { tr += gre[bd]; ti_ += gim[bd]; }
                        const int bc = A.pair_conj[(j0 + q) * MF_NA + i];
                        if (bc >= 0) { tr += gre[bc]; ti_ -= gim[bc]; }
                        const float s_ = 0.5f * (gr[q] + tr), a_ = ti_ - gi[q];
                        gr[q] = s_; gi[q] = -a_;                 // the "-Gi" plane (ih ^ sign) then holds +A
                    }
                }
            }
        }
        uint32_t rh, rl, ih, il;
        split2(gr[0] * gs, gr[1] * gs, rh, rl);
        split2(gi[0] * gs, gi[1] * gs, ih, il);
        const int off = ((((tile * 2 + ks) * 2 + h) * 32 + row) * 4 + jp) * 4;
        *reinterpret_cast<uint32_t*>(g_img + 0 * MB_PLANE + off) = rh;
        *reinterpret_cast<uint32_t*>(g_img + 1 * MB_PLANE + off) = ih;
        *reinterpret_cast<uint32_t*>(g_img + 2 * MB_PLANE + off) = ih ^ 0x80008000u;
        *reinterpret_cast<uint32_t*>(g_img + 3 * MB_PLANE + off) = rl;
        *reinterpret_cast<uint32_t*>(g_img + 4 * MB_PLANE + off) = il;
        *reinterpret_cast<uint32_t*>(g_img + 5 * MB_PLANE + off) = il ^ 0x80008000u;
    }
}

// CPLX: psky is complex (interleaved).  With Z = sum_ij conj(F_ij) g_ij = conj(sum_i conj(E_i) T_i) the two
// gradient planes are Re Z and Im Z: the same T, a second lane-local contraction (Ei Tr - Er Ti) -- the
// imaginary plane costs 32 FMAs per pixel tile and wave instead of a second pass.
// TAMAX: row / column tiles the instantiation is compiled for (round 4).  4: any array of up to 128 antennas.  2: arrays of up to 64
// antennas (HERA-19, HERA-37, the tile shards of a rank) -- four accumulators instead of eight: half the zero-fill moves per
// pixel tile (a quarter of the kernel's VALU instructions at 37 antennas) and 60 registers less; the tile loops of the
// general instantiation only DROP the iterations of absent tiles, they still pay for their registers.
template <bool CPLX, int TAMAX>
__global__ void __launch_bounds__(512, 2)
fringe_ant_bwd_kernel(AntBwdArgs A)
{
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char* g_img = smem;              // planes: 0 Gr_hi, 1 Gi_hi, 2 -Gi_hi, 3 Gr_lo, 4 Gi_lo, 5 -Gi_lo
    double* ant_lds = reinterpret_cast<double*>(smem + 6 * MB_PLANE);      // [128][3]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // 1-D grid, channel fastest: blocks that run together share (t, split), i.e. the same pointing
    // vectors (L2 hits), and no grid dimension hits the 65535 cap
    const int f = blockIdx.x % A.Nf, ts = blockIdx.x / A.Nf;
    const int t = ts / A.S, split = ts % A.S;
    const int TA = (A.Nant + 31) / 32;

    float gs;
    bwd_stage_diag<CPLX, 512>(A, g_img, ant_lds, t, f, TA, gs);
    __syncthreads();

    const double* sd = A.sdir + (size_t)t * 3 * A.Pstride;
    float* orow = A.gpsky + (size_t)t * A.st_t + (size_t)f * A.st_f;
    const float inv = 1.0f / gs;
    const int h = lane >> 5;
    const int ntile = A.Pstride / 32;
    const int tbeg = split * A.tiles_per_split;
    const int tend = min(ntile, tbeg + A.tiles_per_split);

    uint32_t gl0 = (h * 32 + (lane & 31)) * 16, gl1 = gl0 + 3 * MB_PLANE;
    asm volatile("" : "+v"(gl1));                     // opaque: keeps gl1 a second base register
    for (int pt = tbeg + wave; pt < tend; pt += 8) {
        const int p = pt * 32 + (lane & 31);
        const double sx = sd[p], sy = sd[A.Pstride + p], sz = sd[2 * (size_t)A.Pstride + p];
        f32x16 accR[TAMAX], accI[TAMAX];
#pragma unroll
        for (int q = 0; q < TAMAX; ++q)
#pragma unroll
            for (int e = 0; e < 16; ++e) { accR[q][e] = 0.f; accI[q][e] = 0.f; }
        float part = 0.f, parti = 0.f;
#pragma unroll
        for (int tjr = 0; tjr < TAMAX; ++tjr) {
            const int tj = TAMAX - 1 - tjr;                  // descending: row tile tj completes here
            if (tj < TA) {
                float ec[16], es[16];                        // E of antennas 32 tj + (e&3) + 8 (e>>2) + 4 h
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    if (32 * tj + 16 * ks >= A.Nant) {           // uniform: 16 padding antennas (G columns and rows are zero)
#pragma unroll
                        for (int jj = 0; jj < 8; ++jj) { ec[8 * ks + jj] = 0.f; es[8 * ks + jj] = 0.f; }
                        continue;
                    }
                    uint4 Erh, Erl, Eih, Eil;
                    uint32_t* erh = reinterpret_cast<uint32_t*>(&Erh); uint32_t* erl = reinterpret_cast<uint32_t*>(&Erl);
                    uint32_t* eih = reinterpret_cast<uint32_t*>(&Eih); uint32_t* eil = reinterpret_cast<uint32_t*>(&Eil);
                    // mirror group (round 5, MIRROR PAIRS at the top of the file): the second octet of this K step holds the mirror
                    // antennas of the first -- conjugate phasors, not evaluated again, and their f16 halves are those of the
                    // first octet with the imaginary plane's sign bits flipped (uniform branch)
                    const bool mir = (A.mirror >> (2 * tj + ks)) & 1;
#pragma unroll
                    for (int jq = 0; jq < 2; ++jq) {
                        // the 8 antennas of this half K step (4 per half wave) are all padding: uniform skip -- 19 antennas
                        // generate 24 phasors per pixel instead of 32, 37 generate 40 instead of 48 (their G rows and columns
                        // are zero, so E = 0 gives the same bits)
                        if (32 * tj + 16 * ks + 8 * jq >= A.Nant) {
#pragma unroll
                            for (int u = 0; u < 4; ++u) { ec[8 * ks + 4 * jq + u] = 0.f; es[8 * ks + 4 * jq + u] = 0.f; }
                            continue;
                        }
                        if (jq == 1 && mir) {
#pragma unroll
                            for (int u = 0; u < 4; ++u) { ec[8 * ks + 4 + u] = ec[8 * ks + u]; es[8 * ks + 4 + u] = -es[8 * ks + u]; }
                            continue;
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int jj = 4 * jq + u;
                            const int an = 32 * tj + (jj & 3) + 8 * (2 * ks + (jj >> 2)) + 4 * h;
                            const double ph = phase3(ant_lds[3 * an], sx, ant_lds[3 * an + 1], sy, ant_lds[3 * an + 2], sz);
                            const float rr = turn_frac(ph);
                            ec[8 * ks + jj] = __builtin_amdgcn_cosf(rr);
                            es[8 * ks + jj] = __builtin_amdgcn_sinf(rr);
                        }
                    }
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        split2_plain(ec[8 * ks + 2 * q], ec[8 * ks + 2 * q + 1], erh[q], erl[q]);
                        split2_plain(es[8 * ks + 2 * q], es[8 * ks + 2 * q + 1], eih[q], eil[q]);
                    }
                    if (mir && 32 * tj + 16 * ks + 8 < A.Nant) {
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            erh[q + 2] = erh[q]; erl[q + 2] = erl[q];
                            eih[q + 2] = eih[q] ^ 0x80008000u; eil[q + 2] = eil[q] ^ 0x80008000u;
                        }
                    } else {
#pragma unroll
                        for (int q = 2; q < 4; ++q) {
                            split2_plain(ec[8 * ks + 2 * q], ec[8 * ks + 2 * q + 1], erh[q], erl[q]);
                            split2_plain(es[8 * ks + 2 * q], es[8 * ks + 2 * q + 1], eih[q], eil[q]);
                        }
                    }
#pragma unroll
                    for (int ti = 0; ti < TAMAX; ++ti) {
                        if (ti <= tj) {
                            // two lane bases + 16-bit immediates reach all six planes (left alone, the compiler
                            // keeps 19 per-tile bases and adds the plane offsets: 62 v_add per pixel tile)
                            const int tk = (tri_index(ti, tj) * 2 + ks) * 1024;
                            if constexpr (!CPLX) {
                                if (ti == tj) {              // (compile-time after unrolling) symmetric form: 9 MFMAs
                                    const uint4 Gnh = *reinterpret_cast<const uint4*>(g_img + gl0 + 2 * MB_PLANE + tk);
                                    const uint4 Gnl = *reinterpret_cast<const uint4*>(g_img + gl1 + 2 * MB_PLANE + tk);
                                    const uint4 Grh = *reinterpret_cast<const uint4*>(g_img + gl0 + 0 * MB_PLANE + tk);
                                    const uint4 Grl = *reinterpret_cast<const uint4*>(g_img + gl1 + 0 * MB_PLANE + tk);
                                    accR[ti] = RIME_MFMA(Grh, Erh, accR[ti]);
                                    accI[ti] = RIME_MFMA(Grh, Eih, accI[ti]);
                                    accI[ti] = RIME_MFMA(Gnh, Erh, accI[ti]);
                                    accR[ti] = RIME_MFMA(Grh, Erl, accR[ti]);
                                    accI[ti] = RIME_MFMA(Grh, Eil, accI[ti]);
                                    accI[ti] = RIME_MFMA(Gnh, Erl, accI[ti]);
                                    accR[ti] = RIME_MFMA(Grl, Erh, accR[ti]);
                                    accI[ti] = RIME_MFMA(Grl, Eih, accI[ti]);
                                    accI[ti] = RIME_MFMA(Gnl, Erh, accI[ti]);
                                    continue;
                                }
                            }
                            const uint4 Grh = *reinterpret_cast<const uint4*>(g_img + gl0 + 0 * MB_PLANE + tk);
                            const uint4 Gih = *reinterpret_cast<const uint4*>(g_img + gl0 + 1 * MB_PLANE + tk);
                            const uint4 Gnh = *reinterpret_cast<const uint4*>(g_img + gl0 + 2 * MB_PLANE + tk);
                            const uint4 Grl = *reinterpret_cast<const uint4*>(g_img + gl1 + 0 * MB_PLANE + tk);
                            const uint4 Gil = *reinterpret_cast<const uint4*>(g_img + gl1 + 1 * MB_PLANE + tk);
                            const uint4 Gnl = *reinterpret_cast<const uint4*>(g_img + gl1 + 2 * MB_PLANE + tk);
                            accR[ti] = RIME_MFMA(Grh, Erh, accR[ti]);
                            accI[ti] = RIME_MFMA(Grh, Eih, accI[ti]);
                            accR[ti] = RIME_MFMA(Gih, Eih, accR[ti]);
                            accI[ti] = RIME_MFMA(Gnh, Erh, accI[ti]);
                            accR[ti] = RIME_MFMA(Grh, Erl, accR[ti]);
                            accI[ti] = RIME_MFMA(Grh, Eil, accI[ti]);
                            accR[ti] = RIME_MFMA(Gih, Eil, accR[ti]);
                            accI[ti] = RIME_MFMA(Gnh, Erl, accI[ti]);
                            accR[ti] = RIME_MFMA(Grl, Erh, accR[ti]);
                            accI[ti] = RIME_MFMA(Grl, Eih, accI[ti]);
                            accR[ti] = RIME_MFMA(Gil, Eih, accR[ti]);
                            accI[ti] = RIME_MFMA(Gnl, Erh, accI[ti]);
                        }
                    }
                }
                // row tile tj is complete: contract with E_i of the same antennas (lane-local)
                RIME_MFMA_SETTLE();                          // see rime_common.h: packed readers of fresh accumulators
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    part = fmaf(ec[e], accR[tj][e], part);
                    part = fmaf(es[e], accI[tj][e], part);
                }
                if constexpr (CPLX) {
                    // the imaginary plane in chains of its own (fused with `part` the compiler emits v_pk_fma_f32)
                    float pa = 0.f, pb = 0.f;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        pa = fmaf(es[e], accR[tj][e], pa);
                        pb = fmaf(ec[e], accI[tj][e], pb);
                    }
                    parti += pa - pb;
                }
            }
        }
        part += __shfl_xor(part, 32, 64);
        if constexpr (CPLX) parti += __shfl_xor(parti, 32, 64);
        if (h == 0) {
            float* o = orow + (size_t)p * A.st_p;
            if constexpr (CPLX) {
                const float im = parti * inv * A.imsign;
                float2* o2 = reinterpret_cast<float2*>(o);
                *o2 = A.accumulate ? make_float2(o2->x + part * inv, o2->y + im) : make_float2(part * inv, im);
            } else {
                *o = A.accumulate ? *o + part * inv : part * inv;
            }
        }
    }
}


// Cross block of the backward (arrays with more than 128 antennas): rows i in group I (antpos rows
// 0..127), columns j in group J (rows 128..255), all 16 tiles.  T_i = sum_j conj(G[i,j]) E_j as in
// the diagonal kernel; the E_i of the final contraction belong to the other group and are
// generated after the MFMA loop.  Four G planes (Gr, Gi; hi, lo; 128 KB) -- the -Gi products use
// a sign-flipped Er fragment instead (8 v_xor per 16 antennas, shared by the 4 row tiles).
// Adds to gpsky (A.accumulate): every pixel is owned by one lane, launches are stream-ordered.
// Shapes: group I = A.rows_i antpos rows (TI = rows_i / 32 row tiles), group J = the remaining rows (TJ tiles);
// tiles keep the 4 x 4 numbering of the full shape.
constexpr int MX_PLANE = 16 * 2 * 2 * 32 * 16;        // 32768 B
constexpr size_t MX_LDS = 4 * (size_t)MX_PLANE + 2 * MF_NA * 3 * sizeof(double);

template <bool CPLX>
__global__ void __launch_bounds__(512, 2)
fringe_ant_bwd_cross_kernel(AntBwdArgs A)
{
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char* g_img = smem;                       // planes: 0 Gr_hi, 1 Gi_hi, 2 Gr_lo, 3 Gi_lo
    double* ant_lds = reinterpret_cast<double*>(smem + 4 * MX_PLANE);      // [256][3]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // 1-D grid, channel fastest: blocks that run together share (t, split), i.e. the same pointing
    // vectors (L2 hits), and no grid dimension hits the 65535 cap
    const int f = blockIdx.x % A.Nf, ts = blockIdx.x / A.Nf;
    const int t = ts / A.S, split = ts % A.S;
    const int TI = A.rows_i / 32, TJ = (A.Nant - A.rows_i) / 32, JB = A.rows_i;      // uniform

    const double nu_c = A.sign * A.freqs[f] * (1.0 / 2.99792458e8);
    for (int i = tid; i < 2 * MF_NA * 3; i += 512)
        ant_lds[i] = (i < A.Nant * 3) ? nu_c * A.antpos[i] : 0.0;

    const float gs = A.gscale[t * A.Nf + f];
    const float* gre = A.gvt + ((size_t)t * A.Nf + f) * 2 * A.Nbl;
    const float* gim = gre + A.Nbl;
    for (int e = tid; e < 16 * 2 * 2 * 32 * 4; e += 512) {
        const int jp = e & 3, row = (e >> 2) & 31, h = (e >> 7) & 1, ks = (e >> 8) & 1, tile = e >> 9;
        const int ti = tile >> 2, tj = tile & 3;
        if (ti >= TI || tj >= TJ) continue;               // tiles outside the block's shape are never read
        const int i = 32 * ti + row;
        const int j0 = 32 * tj + ((2 * jp) & 3) + 8 * (2 * ks + (jp >> 1)) + 4 * h;
        float gr[2] = {0.f, 0.f}, gi[2] = {0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int bd = A.pair_direct[i * MF_NA + j0 + q];
            if (bd >= 0) { gr[q] += gre[bd]; gi[q] += gim[bd]; }
            const int bc = A.pair_conj[i * MF_NA + j0 + q];
            if (bc >= 0) { gr[q] += gre[bc]; gi[q] -= gim[bc]; }
        }
        uint32_t rh, rl, ih, il;
        split2(gr[0] * gs, gr[1] * gs, rh, rl);
        split2(gi[0] * gs, gi[1] * gs, ih, il);
        const int off = ((((tile * 2 + ks) * 2 + h) * 32 + row) * 4 + jp) * 4;
        *reinterpret_cast<uint32_t*>(g_img + 0 * MX_PLANE + off) = rh;
        *reinterpret_cast<uint32_t*>(g_img + 1 * MX_PLANE + off) = ih;
        *reinterpret_cast<uint32_t*>(g_img + 2 * MX_PLANE + off) = rl;
        *reinterpret_cast<uint32_t*>(g_img + 3 * MX_PLANE + off) = il;
    }
    __syncthreads();

    const double* sd = A.sdir + (size_t)t * 3 * A.Pstride;
    float* orow = A.gpsky + (size_t)t * A.st_t + (size_t)f * A.st_f;
    const float inv = 1.0f / gs;
    const int h = lane >> 5;
    const int ntile = A.Pstride / 32;
    const int tbeg = split * A.tiles_per_split;
    const int tend = min(ntile, tbeg + A.tiles_per_split);

    for (int pt = tbeg + wave; pt < tend; pt += 8) {
        const int p = pt * 32 + (lane & 31);
        const double sx = sd[p], sy = sd[A.Pstride + p], sz = sd[2 * (size_t)A.Pstride + p];
        f32x16 accR[4], accI[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 16; ++e) { accR[q][e] = 0.f; accI[q][e] = 0.f; }
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) {
            if (tj >= TJ) continue;                                     // uniform
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                uint4 Erh, Erl, Eih, Eil, Nrh, Nrl;
                uint32_t* erh = reinterpret_cast<uint32_t*>(&Erh); uint32_t* erl = reinterpret_cast<uint32_t*>(&Erl);
                uint32_t* eih = reinterpret_cast<uint32_t*>(&Eih); uint32_t* eil = reinterpret_cast<uint32_t*>(&Eil);
                float ec[8], es[8];
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const int an = JB + 32 * tj + (jj & 3) + 8 * (2 * ks + (jj >> 2)) + 4 * h;
                    const double ph = phase3(ant_lds[3 * an], sx, ant_lds[3 * an + 1], sy, ant_lds[3 * an + 2], sz);
                    const float rr = turn_frac(ph);
                    ec[jj] = __builtin_amdgcn_cosf(rr);
                    es[jj] = __builtin_amdgcn_sinf(rr);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    split2_plain(ec[2 * q], ec[2 * q + 1], erh[q], erl[q]);
                    split2_plain(es[2 * q], es[2 * q + 1], eih[q], eil[q]);
                }
                Nrh = make_uint4(Erh.x ^ 0x80008000u, Erh.y ^ 0x80008000u, Erh.z ^ 0x80008000u, Erh.w ^ 0x80008000u);
                Nrl = make_uint4(Erl.x ^ 0x80008000u, Erl.y ^ 0x80008000u, Erl.z ^ 0x80008000u, Erl.w ^ 0x80008000u);
#pragma unroll
                for (int ti = 0; ti < 4; ++ti) {
                    if (ti >= TI) continue;                             // uniform
                    const int off = ((((ti * 4 + tj) * 2 + ks) * 2 + h) * 32 + (lane & 31)) * 16;
                    const uint4 Grh = *reinterpret_cast<const uint4*>(g_img + 0 * MX_PLANE + off);
                    const uint4 Gih = *reinterpret_cast<const uint4*>(g_img + 1 * MX_PLANE + off);
                    const uint4 Grl = *reinterpret_cast<const uint4*>(g_img + 2 * MX_PLANE + off);
                    const uint4 Gil = *reinterpret_cast<const uint4*>(g_img + 3 * MX_PLANE + off);
                    accR[ti] = RIME_MFMA(Grh, Erh, accR[ti]);
                    accI[ti] = RIME_MFMA(Grh, Eih, accI[ti]);
                    accR[ti] = RIME_MFMA(Gih, Eih, accR[ti]);
                    accI[ti] = RIME_MFMA(Gih, Nrh, accI[ti]);
                    accR[ti] = RIME_MFMA(Grh, Erl, accR[ti]);
                    accI[ti] = RIME_MFMA(Grh, Eil, accI[ti]);
                    accR[ti] = RIME_MFMA(Gih, Eil, accR[ti]);
                    accI[ti] = RIME_MFMA(Gih, Nrl, accI[ti]);
                    accR[ti] = RIME_MFMA(Grl, Erh, accR[ti]);
                    accI[ti] = RIME_MFMA(Grl, Eih, accI[ti]);
                    accR[ti] = RIME_MFMA(Gil, Eih, accR[ti]);
                    accI[ti] = RIME_MFMA(Gil, Nrh, accI[ti]);
                }
                __builtin_amdgcn_sched_barrier(0);      // keep the live ranges of one K step apart
            }
        }
        // contraction with E_i of group I: the D rows this lane holds.  As in the diagonal kernel: a margin before
        // the first accumulator read, and the imaginary plane in chains of its own (pa, pb) so that the compiler
        // has no (part, parti) pair to fuse into packed f32 operations (rime_common.h)
        RIME_MFMA_SETTLE();
        float part = 0.f, parti = 0.f;
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) {
            if (ti >= TI) continue;                                     // uniform
            float pa = 0.f, pb = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int an = 32 * ti + (e & 3) + 8 * (e >> 2) + 4 * h;
                const double ph = phase3(ant_lds[3 * an], sx, ant_lds[3 * an + 1], sy, ant_lds[3 * an + 2], sz);
                const float rr = turn_frac(ph);
                const float ce = __builtin_amdgcn_cosf(rr), se = __builtin_amdgcn_sinf(rr);
                part = fmaf(ce, accR[ti][e], part);
                part = fmaf(se, accI[ti][e], part);
                if constexpr (CPLX) {
                    pa = fmaf(se, accR[ti][e], pa);
                    pb = fmaf(ce, accI[ti][e], pb);
                }
                if ((e & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (CPLX) parti += pa - pb;
        }
        part += __shfl_xor(part, 32, 64);
        if constexpr (CPLX) parti += __shfl_xor(parti, 32, 64);
        if (h == 0) {
            float* o = orow + (size_t)p * A.st_p;
            if constexpr (CPLX) {
                const float im = parti * inv * A.imsign;
                float2* o2 = reinterpret_cast<float2*>(o);
                *o2 = A.accumulate ? make_float2(o2->x + part * inv, o2->y + im) : make_float2(part * inv, im);
            } else {
                *o = A.accumulate ? *o + part * inv : part * inv;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// CONJUGATE-PAIR FORM, backward (round 5; forward: fringe_pair_fwd_kernel).  With e = c + i s the phasors of the F firsts,
//     gpsky[p] = Re sum_ij ( a_ij conj(e_i) e_j + b_ij e_i e_j ),     a = conj(G_A),  b = conj(G_B)
// where G_A[i,j] / G_B[i,j] collect the visibility gradients of the baselines that received A[i,j] / B[i,j] (and their
// conjugates) in the forward -- the four pairs {i, j}, {i', j'}, {i', j}, {i, j'} of two mirror pairs in ONE complex entry
// each.  In real planes
//     gpsky = sum_i c_i ( N1 c + N3 s )_i + s_i ( N2 s + N4 c )_i,   N1 = ar + br,  N2 = ar - br,  N3 = -(ai + bi),  N4 = ai - bi
// block-upper-triangular like a and b, so this is the generic diagonal backward at HALF the rows -- 64 phasors per pixel and
// 12 + 9 + 9 MFMAs per 16-antenna K step pair (four planes on the off-diagonal tile; on a diagonal tile N1, N2 are
// symmetrised and c^T N3 s + s^T N4 c = c^T (N3 + N4^T) s: three) -- 60 MFMAs per 32 pixels instead of 216 for 128 antennas.
// The hub (CEN, see the forward) adds sum_j u_j c_j + v_j s_j with two real vectors: they are the INITIAL values of the
// accumulators.  G is scaled by gscale / 8: an entry sums up to eight gradients.
// ---------------------------------------------------------------------------------------
struct PairBwdArgs : AntBwdArgs {
    const int* centre;         // CEN: [2][128] slots of the hub's baselines (as PairArgs::centre)
};

// TF: row tiles the instantiation is compiled for -- 2: up to 64 rows, tiles (0,0) (0,1) (1,1); 1: up to 32 rows (HERA-37-class
// arrays), one tile: half the accumulators, a third of the planes
template <int TF> constexpr int pb_plane() { return (TF == 1 ? 1 : 3) * 2 * 2 * 32 * 16; }   // bytes per plane: (tile, ks, h, row) x 8 f16
// four waves per block and <= 168 registers: THREE blocks per CU (51 KB of LDS each) -- 12 % faster than one block of eight waves
// at 172 registers (profiles/r05/pair_form.txt): with a third of the matrix work of the generic kernel a wave waits more often
// for its own phasors, and only other waves fill those slots
constexpr int PB_THREADS = 256;
template <int TF> constexpr size_t pb_lds() { return 8 * (size_t)pb_plane<TF>() + 64 * 3 * sizeof(double) + 2 * 64 * sizeof(float); }

template <bool CEN, bool FLAT, int TF>
__global__ void __launch_bounds__(PB_THREADS, 3)
fringe_pair_bwd_kernel(PairBwdArgs A)
{
    constexpr int PB_PLANE = pb_plane<TF>();
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char* g_img = smem;              // planes: 0 N1, 1 N3, 2 N2, 3 N4 (hi), 4..7 the same (lo)
    double* ant_lds = reinterpret_cast<double*>(smem + 8 * PB_PLANE);      // [64][3]
    float* cen_u = reinterpret_cast<float*>(smem + 8 * PB_PLANE + 64 * 3 * sizeof(double));
    float* cen_v = cen_u + 64;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int f = blockIdx.x % A.Nf, ts = blockIdx.x / A.Nf;
    const int t = ts / A.S, split = ts % A.S;
    const int TA = (A.Nant + 31) / 32;

    // ---- staging: antenna coordinates (x sign nu / c), the eight planes in A-fragment order, the hub's vectors
    const double nu_c = A.sign * A.freqs[f] * (1.0 / 2.99792458e8);
    for (int i = tid; i < 64 * 3; i += PB_THREADS) ant_lds[i] = (i < A.Nant * 3) ? nu_c * A.antpos[i] : 0.0;
    const float gs = A.gscale[t * A.Nf + f] * 0.125f;
    const float* gre = A.gvt + ((size_t)t * A.Nf + f) * 2 * A.Nbl;
    const float* gim = gre + A.Nbl;
    // gradient with respect to V[r, c] of the virtual 128-row block: direct slot + conjugate of the conj slot
    auto grad_of = [&](int r, int c, float& vr, float& vi) {
        const int bd = A.pair_direct[r * MF_NA + c];
        if (bd >= 0) { vr += gre[bd]; keep_scalar(vr); vi += gim[bd]; keep_scalar(vi); }      // (scalar adds: see keep_scalar)
        const int bc = A.pair_conj[r * MF_NA + c];
        if (bc >= 0) { vr += gre[bc]; keep_scalar(vr); vi -= gim[bc]; keep_scalar(vi); }
    };
    // a[i,j] = conj(g V[i,j]) + g V[i',j'];  b[i,j] = g V[j,i'] (+ g V[i,j'] on the off-diagonal tile): what the forward wrote where
    auto entry = [&](int i, int j, bool off, float& ar, float& ai, float& br, float& bi) {
        float xr = 0.f, xi = 0.f, yr = 0.f, yi = 0.f;
        grad_of(i, j, xr, xi);
        grad_of(64 + i, 64 + j, yr, yi);
        ar = xr + yr; keep_scalar(ar); ai = yi - xi;
        br = 0.f; bi = 0.f;
        grad_of(j, 64 + i, br, bi);
        if (off) grad_of(i, 64 + j, br, bi);
    };
    for (int e = tid; e < (TF == 1 ? 1 : 3) * 2 * 2 * 32 * 4; e += PB_THREADS) {
        const int jp = e & 3, row = (e >> 2) & 31, h = (e >> 7) & 1, ks = (e >> 8) & 1, tile = e >> 9;
        const int ti = tile == 2 ? 1 : 0, tj = tile == 0 ? 0 : 1;
        const int i = 32 * ti + row;
        const int j0 = 32 * tj + ((2 * jp) & 3) + 8 * (2 * ks + (jp >> 1)) + 4 * h;
        float n1[2] = {0.f, 0.f}, n2[2] = {0.f, 0.f}, n3[2] = {0.f, 0.f}, n4[2] = {0.f, 0.f};
        if (tj < TA) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int j = j0 + q;
                float ar, ai, br, bi;
                entry(i, j, ti != tj, ar, ai, br, bi);
                if (ti != tj) {
                    n1[q] = ar + br; keep_scalar(n1[q]); n2[q] = ar - br; keep_scalar(n2[q]);
                    n3[q] = -(ai + bi); keep_scalar(n3[q]); n4[q] = ai - bi;
                } else {
                    float tr, ti_, ur, ui;                        // the (j, i) element of the same tile
                    entry(j, i, false, tr, ti_, ur, ui);
                    float s1 = ar + br, s2 = tr + ur, d1 = ar - br, d2 = tr - ur;
                    keep_scalar(s1); keep_scalar(s2); keep_scalar(d1);
                    n1[q] = 0.5f * (s1 + s2); keep_scalar(n1[q]);
                    n2[q] = 0.5f * (d1 + d2); keep_scalar(n2[q]);
                    float e1 = ti_ - ui, e2 = ai + bi;
                    keep_scalar(e1);
                    n3[q] = e1 - e2;                              // N3[i,j] + N4[j,i]
                }
                keep_scalar(n1[q]); keep_scalar(n2[q]); keep_scalar(n3[q]); keep_scalar(n4[q]);
            }
        }
        uint32_t hi_, lo_;
        const int off = ((((tile * 2 + ks) * 2 + h) * 32 + row) * 4 + jp) * 4;
        split2(n1[0] * gs, n1[1] * gs, hi_, lo_);
        *reinterpret_cast<uint32_t*>(g_img + 0 * PB_PLANE + off) = hi_; *reinterpret_cast<uint32_t*>(g_img + 4 * PB_PLANE + off) = lo_;
        split2(n3[0] * gs, n3[1] * gs, hi_, lo_);
        *reinterpret_cast<uint32_t*>(g_img + 1 * PB_PLANE + off) = hi_; *reinterpret_cast<uint32_t*>(g_img + 5 * PB_PLANE + off) = lo_;
        split2(n2[0] * gs, n2[1] * gs, hi_, lo_);
        *reinterpret_cast<uint32_t*>(g_img + 2 * PB_PLANE + off) = hi_; *reinterpret_cast<uint32_t*>(g_img + 6 * PB_PLANE + off) = lo_;
        split2(n4[0] * gs, n4[1] * gs, hi_, lo_);
        *reinterpret_cast<uint32_t*>(g_img + 3 * PB_PLANE + off) = hi_; *reinterpret_cast<uint32_t*>(g_img + 7 * PB_PLANE + off) = lo_;
    }
    if constexpr (CEN) {
        if (tid < 64) {
            // Re(conj(g) V[c, j]) with V[c, j] = e_j (row j) and conj(e_j) (row 64 + j): u c_j + v s_j
            float u = 0.f, v = 0.f;
            int b = A.centre[tid];
            if (b >= 0) { u += gre[b]; keep_scalar(u); v += gim[b]; keep_scalar(v); }
            b = A.centre[MF_NA + tid];
            if (b >= 0) { u += gre[b]; keep_scalar(u); v -= gim[b]; keep_scalar(v); }
            b = A.centre[64 + tid];
            if (b >= 0) { u += gre[b]; keep_scalar(u); v -= gim[b]; keep_scalar(v); }
            b = A.centre[MF_NA + 64 + tid];
            if (b >= 0) { u += gre[b]; keep_scalar(u); v += gim[b]; keep_scalar(v); }
            u *= gs; keep_scalar(u); v *= gs;
            cen_u[tid] = u; cen_v[tid] = v;
        }
    }
    __syncthreads();

    const double* sd = A.sdir + (size_t)t * 3 * A.Pstride;
    float* orow = A.gpsky + (size_t)t * A.st_t + (size_t)f * A.st_f;
    const float inv = 1.0f / gs;
    const int h = lane >> 5;
    const int ntile = A.Pstride / 32;
    const int tbeg = split * A.tiles_per_split;
    const int tend = min(ntile, tbeg + A.tiles_per_split);

    const uint32_t gl0 = (h * 32 + (lane & 31)) * 16;    // one lane base + 16-bit immediates reach all eight planes (48 KB)
    for (int pt = tbeg + wave; pt < tend; pt += PB_THREADS / 64) {
        const int p = pt * 32 + (lane & 31);
        const double sx = sd[p], sy = sd[A.Pstride + p], sz = FLAT ? 0.0 : sd[2 * (size_t)A.Pstride + p];
        f32x16 accR[TF], accI[TF];
#pragma unroll
        for (int q = 0; q < TF; ++q)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                if constexpr (CEN) {
                    accR[q][e] = cen_u[32 * q + (e & 3) + 8 * (e >> 2) + 4 * h];
                    accI[q][e] = cen_v[32 * q + (e & 3) + 8 * (e >> 2) + 4 * h];
                } else { accR[q][e] = 0.f; accI[q][e] = 0.f; }
            }
        float part = 0.f;
#pragma unroll
        for (int tjr = 0; tjr < TF; ++tjr) {
            const int tj = TF - 1 - tjr;                          // descending: row tile tj completes here
            if (tj < TA) {
                float ec[16], es[16];                        // E of antennas 32 tj + (e&3) + 8 (e>>2) + 4 h
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    if (32 * tj + 16 * ks >= A.Nant) {           // uniform: 16 padding rows (their G rows and columns are zero)
#pragma unroll
                        for (int jj = 0; jj < 8; ++jj) { ec[8 * ks + jj] = 0.f; es[8 * ks + jj] = 0.f; }
                        continue;
                    }
                    uint4 Erh, Erl, Eih, Eil;
                    uint32_t* erh = reinterpret_cast<uint32_t*>(&Erh); uint32_t* erl = reinterpret_cast<uint32_t*>(&Erl);
                    uint32_t* eih = reinterpret_cast<uint32_t*>(&Eih); uint32_t* eil = reinterpret_cast<uint32_t*>(&Eil);
#pragma unroll
                    for (int jq = 0; jq < 2; ++jq) {
                        if (32 * tj + 16 * ks + 8 * jq >= A.Nant) {  // uniform: 8 padding rows
#pragma unroll
                            for (int u = 0; u < 4; ++u) { ec[8 * ks + 4 * jq + u] = 0.f; es[8 * ks + 4 * jq + u] = 0.f; }
                            continue;
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int jj = 4 * jq + u;
                            const int an = 32 * tj + (jj & 3) + 8 * (2 * ks + (jj >> 2)) + 4 * h;
                            const double ph = phase_of<FLAT>(ant_lds[3 * an], sx, ant_lds[3 * an + 1], sy, ant_lds[3 * an + 2], sz);
                            const float rr = turn_frac(ph);
                            ec[8 * ks + jj] = __builtin_amdgcn_cosf(rr);
                            es[8 * ks + jj] = __builtin_amdgcn_sinf(rr);
                        }
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        split2_plain(ec[8 * ks + 2 * q], ec[8 * ks + 2 * q + 1], erh[q], erl[q]);
                        split2_plain(es[8 * ks + 2 * q], es[8 * ks + 2 * q + 1], eih[q], eil[q]);
                    }
#pragma unroll
                    for (int ti = 0; ti < TF; ++ti) {
                        if (ti <= tj) {
                            const int tk = ((ti + tj) * 2 + ks) * 1024;
                            const uint4 N1h = *reinterpret_cast<const uint4*>(g_img + gl0 + 0 * PB_PLANE + tk);
                            const uint4 N3h = *reinterpret_cast<const uint4*>(g_img + gl0 + 1 * PB_PLANE + tk);
                            const uint4 N2h = *reinterpret_cast<const uint4*>(g_img + gl0 + 2 * PB_PLANE + tk);
                            const uint4 N1l = *reinterpret_cast<const uint4*>(g_img + gl0 + 4 * PB_PLANE + tk);
                            const uint4 N3l = *reinterpret_cast<const uint4*>(g_img + gl0 + 5 * PB_PLANE + tk);
                            const uint4 N2l = *reinterpret_cast<const uint4*>(g_img + gl0 + 6 * PB_PLANE + tk);
                            if (ti == tj) {                  // (compile-time after unrolling) diagonal tile: three planes
                                accR[ti] = RIME_MFMA(N1h, Erh, accR[ti]);
                                accI[ti] = RIME_MFMA(N2h, Eih, accI[ti]);
                                accR[ti] = RIME_MFMA(N3h, Eih, accR[ti]);
                                accI[ti] = RIME_MFMA(N2h, Eil, accI[ti]);
                                accR[ti] = RIME_MFMA(N1h, Erl, accR[ti]);
                                accI[ti] = RIME_MFMA(N2l, Eih, accI[ti]);
                                accR[ti] = RIME_MFMA(N3h, Eil, accR[ti]);
                                accR[ti] = RIME_MFMA(N1l, Erh, accR[ti]);
                                accR[ti] = RIME_MFMA(N3l, Eih, accR[ti]);
                                continue;
                            }
                            const uint4 N4h = *reinterpret_cast<const uint4*>(g_img + gl0 + 3 * PB_PLANE + tk);
                            const uint4 N4l = *reinterpret_cast<const uint4*>(g_img + gl0 + 7 * PB_PLANE + tk);
                            accR[ti] = RIME_MFMA(N1h, Erh, accR[ti]);
                            accI[ti] = RIME_MFMA(N2h, Eih, accI[ti]);
                            accR[ti] = RIME_MFMA(N3h, Eih, accR[ti]);
                            accI[ti] = RIME_MFMA(N4h, Erh, accI[ti]);
                            accR[ti] = RIME_MFMA(N1h, Erl, accR[ti]);
                            accI[ti] = RIME_MFMA(N2h, Eil, accI[ti]);
                            accR[ti] = RIME_MFMA(N3h, Eil, accR[ti]);
                            accI[ti] = RIME_MFMA(N4h, Erl, accI[ti]);
                            accR[ti] = RIME_MFMA(N1l, Erh, accR[ti]);
                            accI[ti] = RIME_MFMA(N2l, Eih, accI[ti]);
                            accR[ti] = RIME_MFMA(N3l, Eih, accR[ti]);
                            accI[ti] = RIME_MFMA(N4l, Erh, accI[ti]);
                        }
                    }
                }
                // row tile tj is complete: contract with E_i of the same antennas (lane-local)
                RIME_MFMA_SETTLE();
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    part = fmaf(ec[e], accR[tj][e], part);
                    part = fmaf(es[e], accI[tj][e], part);
                }
            }
        }
        part += __shfl_xor(part, 32, 64);
        if (h == 0) {
            float* o = orow + (size_t)p * A.st_p;
            *o = A.accumulate ? *o + part * inv : part * inv;
        }
    }
}


// Power-of-two pre-scale (and minimum) of the rows the matrix-core kernels contract: scale = 2^floor(log2(2^14 / max|x|))
// (1 for an all-zero row), rowmin = min x.  One block per row; rows are enumerated (d0, d1, d2, d3) in OUTPUT order and
// found through four element strides, so a strided psky view needs no copy.  Replaces a dozen elementwise / reduction
// launches per fringe call (aminmax, maximum, log2, floor, clamp, exp2, where, permute + contiguous) -- host time, not
// device time, is what it saves (a rank's share at 8 GPUs spends a sixth of its step enqueueing).
__global__ void __launch_bounds__(256)
row_scale_kernel(const float* __restrict__ x, int d1, int d2, int d3, long long s0, long long s1, long long s2, long long s3,
                 int L, float* __restrict__ scale, float* __restrict__ rowmin)
{
    __shared__ float wlo[4], whi[4];
    const int r = blockIdx.x, tid = threadIdx.x;
    const int i3 = r % d3, i2 = (r / d3) % d2, i1 = (r / (d3 * d2)) % d1, i0 = r / (d3 * d2 * d1);
    const float* row = x + i0 * s0 + i1 * s1 + i2 * s2 + i3 * s3;
    float lo = INFINITY, hi = -INFINITY;
    if ((L & 3) == 0 && (reinterpret_cast<uintptr_t>(row) & 15) == 0) {
        for (int i = tid; i < L / 4; i += 256) {
            const float4 v = reinterpret_cast<const float4*>(row)[i];
            lo = fminf(fminf(lo, fminf(v.x, v.y)), fminf(v.z, v.w));
            hi = fmaxf(fmaxf(hi, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
        }
    } else {
        for (int i = tid; i < L; i += 256) { const float v = row[i]; lo = fminf(lo, v); hi = fmaxf(hi, v); }
    }
    for (int o = 32; o > 0; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o, 64)); hi = fmaxf(hi, __shfl_xor(hi, o, 64)); }
    if ((tid & 63) == 0) { wlo[tid >> 6] = lo; whi[tid >> 6] = hi; }
    __syncthreads();
    if (tid == 0) {
        lo = fminf(fminf(wlo[0], wlo[1]), fminf(wlo[2], wlo[3]));
        hi = fmaxf(fmaxf(whi[0], whi[1]), fmaxf(whi[2], whi[3]));
        const float amax = fmaxf(hi, -lo);
        float sc = 1.0f;
        if (amax > 0.f && amax < INFINITY) sc = exp2f(fminf(fmaxf(floorf(log2f(16384.0f / amax)), -100.f), 100.f));
        scale[r] = sc;
        if (rowmin) rowmin[r] = lo;
    }
}

// The same for rows of an interleaved COMPLEX psky (full-polarisation layouts): scale from max(|re|, |im|) over the row, and
// the minimum of each plane (the sign-mask-free instantiations of the per-plane passes key on it).  One pass over the row
// instead of torch's abs (a full-size temporary), amax and amin passes: 1.08 -> 0.2 ms per C5 rank step.
__global__ void __launch_bounds__(256)
row_scale_cplx_kernel(const float* __restrict__ x, int d1, int d2, int d3, long long s0, long long s1, long long s2, long long s3,
                      int L, float* __restrict__ scale, float* __restrict__ rowmin_re, float* __restrict__ rowmin_im)
{
    __shared__ float wre[4], wim[4], whi[4];
    const int r = blockIdx.x, tid = threadIdx.x;
    const int i3 = r % d3, i2 = (r / d3) % d2, i1 = (r / (d3 * d2)) % d1, i0 = r / (d3 * d2 * d1);
    const float* row = x + 2 * (i0 * s0 + i1 * s1 + i2 * s2 + i3 * s3);          // strides in complex elements
    float lre = INFINITY, lim = INFINITY, hi = 0.f;
    if ((L & 1) == 0 && (reinterpret_cast<uintptr_t>(row) & 15) == 0) {
        for (int i = tid; i < L / 2; i += 256) {
            const float4 v = reinterpret_cast<const float4*>(row)[i];          // two complex values
            lre = fminf(lre, fminf(v.x, v.z)); lim = fminf(lim, fminf(v.y, v.w));
            hi = fmaxf(hi, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
        }
    } else {
        for (int i = tid; i < L; i += 256) {
            const float a = row[2 * i], b = row[2 * i + 1];
            lre = fminf(lre, a); lim = fminf(lim, b); hi = fmaxf(hi, fmaxf(fabsf(a), fabsf(b)));
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        lre = fminf(lre, __shfl_xor(lre, o, 64)); lim = fminf(lim, __shfl_xor(lim, o, 64)); hi = fmaxf(hi, __shfl_xor(hi, o, 64));
    }
    if ((tid & 63) == 0) { wre[tid >> 6] = lre; wim[tid >> 6] = lim; whi[tid >> 6] = hi; }
    __syncthreads();
    if (tid == 0) {
        lre = fminf(fminf(wre[0], wre[1]), fminf(wre[2], wre[3]));
        lim = fminf(fminf(wim[0], wim[1]), fminf(wim[2], wim[3]));
        hi = fmaxf(fmaxf(whi[0], whi[1]), fmaxf(whi[2], whi[3]));
        float sc = 1.0f;
        if (hi > 0.f && hi < INFINITY) sc = exp2f(fminf(fmaxf(floorf(log2f(16384.0f / hi)), -100.f), 100.f));
        scale[r] = sc;
        if (rowmin_re) rowmin_re[r] = lre;
        if (rowmin_im) rowmin_im[r] = lim;
    }
}

// vis[bl][t][f][c] = sum_s ws[s][t][f][c][bl]: block = (32 baselines, 32 channels, one time); reads are
// coalesced along bl, the 32x32 tile is turned through LDS, writes are 256-B runs along f.
__global__ void __launch_bounds__(256)
reduce_vis_kernel(const float* __restrict__ ws, float* __restrict__ vis, int Nbl, int Nt, int Nf, int S)
{
    __shared__ float tile[32][2][33];
    const int b0 = blockIdx.x * 32, f0 = blockIdx.y * 32, t = blockIdx.z;
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;          // ly: 8 rows
    const size_t slab = (size_t)Nt * Nf * 2 * Nbl;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int f = f0 + ly + 8 * k, bl = b0 + lx;
        float vr = 0.f, vi = 0.f;
        if (f < Nf && bl < Nbl) {
            const float* src = ws + ((size_t)t * Nf + f) * 2 * Nbl + bl;
            for (int s = 0; s < S; ++s) { vr += src[(size_t)s * slab]; vi += src[(size_t)s * slab + Nbl]; }
        }
        tile[ly + 8 * k][0][lx] = vr;
        tile[ly + 8 * k][1][lx] = vi;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int bl = b0 + ly + 8 * k, f = f0 + lx;
        if (bl < Nbl && f < Nf) {
            float2* o = reinterpret_cast<float2*>(vis + (((size_t)bl * Nt + t) * Nf + f) * 2);
            *o = make_float2(tile[lx][0][ly + 8 * k], tile[lx][1][ly + 8 * k]);
        }
    }
}

// gvt[t][f][c][bl] = gvis[bl][t][f][c]: the backward stages G per (t, f) block along baselines
__global__ void __launch_bounds__(256)
transpose_gvis_kernel(const float* __restrict__ gvis, float* __restrict__ gvt, int Nbl, int Nt, int Nf)
{
    __shared__ float tile[32][2][33];
    const int b0 = blockIdx.x * 32, f0 = blockIdx.y * 32, t = blockIdx.z;
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int bl = b0 + ly + 8 * k, f = f0 + lx;
        float2 v = make_float2(0.f, 0.f);
        if (bl < Nbl && f < Nf) v = *reinterpret_cast<const float2*>(gvis + (((size_t)bl * Nt + t) * Nf + f) * 2);
        tile[lx][0][ly + 8 * k] = v.x;
        tile[lx][1][ly + 8 * k] = v.y;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int f = f0 + ly + 8 * k, bl = b0 + lx;
        if (f < Nf && bl < Nbl) {
            float* dst = gvt + ((size_t)t * Nf + f) * 2 * Nbl + bl;
            dst[0] = tile[ly + 8 * k][0][lx];
            dst[Nbl] = tile[ly + 8 * k][1][lx];
        }
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

static void ant_split_plan(int Nt, int Nf, int Pstride, int& S, int& panels_per_split)
{
    S = ant_splits(Nt, Nf, Pstride);
    const int npanel = Pstride / MF_KP;
    panels_per_split = (npanel + S - 1) / S;
    panels_per_split = ((panels_per_split + 3) / 4) * 4;            // splits start on 64-pixel boundaries
    S = (npanel + panels_per_split - 1) / panels_per_split;
}

extern "C" size_t rime_fringe_ant_workspace(int Nbl, int Nt, int Nf, int Pstride)
{
    int S, pps;
    ant_split_plan(Nt, Nf, Pstride, S, pps);
    return (size_t)S * Nbl * Nt * Nf * 2 * sizeof(float);
}

// RIME_FWD_PACKED=0: arrays of 33..48 antennas keep the generic two-tile forward kernel (A/B measurements; ops.py reads
// the same variable for its count of executed MFMAs)
static bool fwd_packed_enabled()
{
    static const int on = [] { const char* e = getenv("RIME_FWD_PACKED"); return e ? atoi(e) : 1; }();
    return on != 0;
}

// RIME_BWD_SMALL=0: arrays of up to 64 antennas keep the four-tile instantiation of the backward kernel (A/B measurements)
static bool bwd_small_enabled()
{
    static const int on = [] { const char* e = getenv("RIME_BWD_SMALL"); return e ? atoi(e) : 1; }();
    return on != 0;
}

static bool cross_shape_ok(int rows_i, int rows_j)
{
    return (rows_i == 32 && rows_j == 32) || (rows_i == 32 && rows_j == 64) || (rows_i == 64 && rows_j == 64) ||
           (rows_i == 128 && rows_j == 128);
}

static bool ant_common_ok(int Nrows, int cross, int Nbl, int Nt, int Nf, int Pstride, long long st_p, int sign, int cplx)
{
    if (st_p != 1 && st_p != 2) return false;
    if (cplx != 0 && (cplx != 1 && cplx != -1)) return false;
    if (cplx != 0 && st_p != 2) return false;                      // complex psky: interleaved (re, im)
    if (cross && cross == Nrows) {                                  // self block: complex psky, 32 / 64 / 96 / 128 rows
        if (cplx == 0 || !(Nrows == 32 || Nrows == 64 || Nrows == 96 || Nrows == 128)) return false;
    } else if (cross ? !cross_shape_ok(cross, Nrows - cross) : (Nrows <= 0 || Nrows > MF_NA)) return false;
    if (Nbl <= 0 || Nt <= 0 || Nt > 65535 || Nf <= 0 || Pstride <= 0 || Pstride % 64 != 0) return false;
    return sign == 1 || sign == -1;
}

template <int TI, int TJ>
static void launch_fwd_cross(const AntArgs& A, dim3 grid, hipStream_t st, bool has_rowmin, int cplx)
{
    using SH = FwdShape<TI, TJ, true>;
    if (cplx) {
        hipLaunchKernelGGL((fringe_ant_fwd_cross_kernel<TI, TJ, false, true>), grid, dim3(SH::NW * 64), SH::LDS, st, A);
        return;
    }
    hipLaunchKernelGGL((fringe_ant_fwd_cross_kernel<TI, TJ, true, false>), grid, dim3(SH::NW * 64), SH::LDS, st, A);
    if (has_rowmin) hipLaunchKernelGGL((fringe_ant_fwd_cross_kernel<TI, TJ, false, false>), grid, dim3(SH::NW * 64), SH::LDS, st, A);
}

extern "C" int rime_fringe_ant_fwd_block(const double* antpos, int Nrows, int cross, int mirror, const double* sdir,
                                         const double* freqs, const float* psky, const float* scale,
                                         const float* rowmin, const int* pair_direct, const int* pair_conj, int Nbl, int Nt, int Nf,
                                         int Pstride, long long st_t, long long st_f, long long st_p, int sign,
                                         int psky_complex, void* workspace, size_t workspace_bytes, void* stream)
{
    if (!antpos || !sdir || !freqs || !psky || !scale || !pair_direct || !pair_conj) return RIME_EINVAL;
    if (!ant_common_ok(Nrows, cross, Nbl, Nt, Nf, Pstride, st_p, sign, psky_complex)) return RIME_EINVAL;
    if (psky_complex && !cross) return RIME_EUNSUPPORTED;          // diagonal blocks: one real plane per call
    AntArgs A{};
    A.antpos = antpos; A.sdir = sdir; A.freqs = freqs; A.psky = psky; A.scale = scale; A.rowmin = psky_complex ? nullptr : rowmin;
    A.pair_direct = pair_direct; A.pair_conj = pair_conj; A.vis = nullptr; A.ws = (float*)workspace;
    A.Nant = Nrows; A.Nbl = Nbl; A.Nt = Nt; A.Nf = Nf; A.Pstride = Pstride;
    A.st_t = st_t; A.st_f = st_f; A.st_p = st_p; A.sign = (double)sign; A.imsign = psky_complex < 0 ? -1.f : 1.f;
    // mirror groups: diagonal blocks on a real plane (the packed 33..48 shape: first row tile only); anything else evaluates
    // every row (the mask is a licence, not an obligation)
    if (mirror < 0 || (Nrows > 0 && Nrows <= MF_NA && (mirror >> ((Nrows + 15) / 16)) != 0)) return RIME_EINVAL;
    A.mirror = (!cross && !psky_complex) ? mirror : 0;
    ant_split_plan(Nt, Nf, Pstride, A.S, A.panels_per_split);
    if (!workspace || workspace_bytes < rime_fringe_ant_workspace(Nbl, Nt, Nf, Pstride)) return RIME_EWORKSPACE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    dim3 grid((unsigned)Nt * A.S * Nf, 1, 1);
    if (cross && cross == Nrows) {
#define RIME_FWD_SELF(TI)                                                                                      \
    do {                                                                                                       \
        using SH = FwdShape<TI, TI, true, true>;                                                               \
        hipLaunchKernelGGL((fringe_ant_fwd_self_kernel<TI>), grid, dim3(SH::NW * 64), SH::LDS, st, A);         \
    } while (0)
        if (Nrows == 32) RIME_FWD_SELF(1);
        else if (Nrows == 64) RIME_FWD_SELF(2);
        else if (Nrows == 96) RIME_FWD_SELF(3);
        else RIME_FWD_SELF(4);
#undef RIME_FWD_SELF
        return check_launch();
    }
    if (cross) {
        const int ti = cross / 32, tj = (Nrows - cross) / 32;
        if (ti == 1 && tj == 1) launch_fwd_cross<1, 1>(A, grid, st, rowmin != nullptr, psky_complex);
        else if (ti == 1 && tj == 2) launch_fwd_cross<1, 2>(A, grid, st, rowmin != nullptr, psky_complex);
        else if (ti == 2 && tj == 2) launch_fwd_cross<2, 2>(A, grid, st, rowmin != nullptr, psky_complex);
        else launch_fwd_cross<4, 4>(A, grid, st, rowmin != nullptr, psky_complex);
        return check_launch();
    }
#define RIME_FWD_PAIR(TA)                                                                                      \
    do {                                                                                                       \
        using SH = FwdShape<TA, TA, false>;                                                                    \
        if (A.mirror) {                                                                                        \
            hipLaunchKernelGGL((fringe_ant_fwd_kernel<TA, true, true>), grid, dim3(SH::NW * 64), SH::LDS, st, A);  \
            if (rowmin) hipLaunchKernelGGL((fringe_ant_fwd_kernel<TA, false, true>), grid, dim3(SH::NW * 64), SH::LDS, st, A); \
            break;                                                                                             \
        }                                                                                                      \
        hipLaunchKernelGGL((fringe_ant_fwd_kernel<TA, true>), grid, dim3(SH::NW * 64), SH::LDS, st, A);        \
        if (rowmin) hipLaunchKernelGGL((fringe_ant_fwd_kernel<TA, false>), grid, dim3(SH::NW * 64), SH::LDS, st, A); \
    } while (0)
    switch ((Nrows + 31) / 32) {
        case 1: RIME_FWD_PAIR(1); break;
        case 2:
            if (Nrows <= 48 && fwd_packed_enabled()) {        // <= 16 antennas in the second row tile: packed planes
                A.mirror &= 3;                                // (no mirror pairs in the packed second row tile)
                if (A.mirror) {
                    hipLaunchKernelGGL((fringe_ant_fwd_packed_kernel<true, true>), grid, dim3(PK::NW * 64), PK::LDS, st, A);
                    if (rowmin) hipLaunchKernelGGL((fringe_ant_fwd_packed_kernel<false, true>), grid, dim3(PK::NW * 64), PK::LDS, st, A);
                    break;
                }
                hipLaunchKernelGGL((fringe_ant_fwd_packed_kernel<true>), grid, dim3(PK::NW * 64), PK::LDS, st, A);
                if (rowmin) hipLaunchKernelGGL((fringe_ant_fwd_packed_kernel<false>), grid, dim3(PK::NW * 64), PK::LDS, st, A);
                break;
            }
            RIME_FWD_PAIR(2);
            break;
        case 3: RIME_FWD_PAIR(3); break;
        default:
            RIME_FWD_PAIR(4);
            break;
    }
#undef RIME_FWD_PAIR
    return check_launch();
}

extern "C" int rime_fringe_row_scale(const float* x, int d0, int d1, int d2, int d3, long long s0, long long s1,
                                     long long s2, long long s3, int L, float* scale, float* rowmin, void* stream)
{
    if (!x || !scale || d0 <= 0 || d1 <= 0 || d2 <= 0 || d3 <= 0 || L <= 0) return RIME_EINVAL;
    const long long rows = (long long)d0 * d1 * d2 * d3;
    if (rows > 0x7fffffffLL) return RIME_EUNSUPPORTED;
    hipLaunchKernelGGL(row_scale_kernel, dim3((unsigned)rows), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       x, d1, d2, d3, s0, s1, s2, s3, L, scale, rowmin);
    return check_launch();
}

extern "C" int rime_fringe_row_scale_cplx(const float* x, int d0, int d1, int d2, int d3, long long s0, long long s1,
                                          long long s2, long long s3, int L, float* scale, float* rowmin_re,
                                          float* rowmin_im, void* stream)
{
    if (!x || !scale || d0 <= 0 || d1 <= 0 || d2 <= 0 || d3 <= 0 || L <= 0) return RIME_EINVAL;
    const long long rows = (long long)d0 * d1 * d2 * d3;
    if (rows > 0x7fffffffLL) return RIME_EUNSUPPORTED;
    hipLaunchKernelGGL(row_scale_cplx_kernel, dim3((unsigned)rows), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       x, d1, d2, d3, s0, s1, s2, s3, L, scale, rowmin_re, rowmin_im);
    return check_launch();
}

extern "C" int rime_fringe_ant_fwd_finish(const void* workspace, size_t workspace_bytes, float* vis,
                                          int Nbl, int Nt, int Nf, int Pstride, void* stream)
{
    if (!vis || Nbl <= 0 || Nt <= 0 || Nf <= 0 || Pstride <= 0 || Pstride % 64 != 0) return RIME_EINVAL;
    if (!workspace || workspace_bytes < rime_fringe_ant_workspace(Nbl, Nt, Nf, Pstride)) return RIME_EWORKSPACE;
    int S, pps;
    ant_split_plan(Nt, Nf, Pstride, S, pps);
    hipLaunchKernelGGL(reduce_vis_kernel, dim3((Nbl + 31) / 32, (Nf + 31) / 32, Nt), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), (const float*)workspace, vis, Nbl, Nt, Nf, S);
    return check_launch();
}

extern "C" int rime_fringe_ant_fwd(const double* antpos, const double* sdir, const double* freqs,
                                   const float* psky, const float* scale, const float* rowmin,
                                   const int* pair_direct,
                                   const int* pair_conj, int Nant, int Nbl, int Nt, int Nf, int Pstride,
                                   long long st_t, long long st_f, long long st_p, int sign, float* vis,
                                   void* workspace, size_t workspace_bytes, void* stream)
{
    if (!vis) return RIME_EINVAL;
    const int rc = rime_fringe_ant_fwd_block(antpos, Nant, 0, 0, sdir, freqs, psky, scale, rowmin, pair_direct, pair_conj,
                                             Nbl, Nt, Nf, Pstride, st_t, st_f, st_p, sign, 0, workspace,
                                             workspace_bytes, stream);
    if (rc != RIME_OK) return rc;
    return rime_fringe_ant_fwd_finish(workspace, workspace_bytes, vis, Nbl, Nt, Nf, Pstride, stream);
}

extern "C" size_t rime_fringe_ant_bwd_workspace(int Nbl, int Nt, int Nf)
{
    return (size_t)Nbl * Nt * Nf * 2 * sizeof(float);
}

extern "C" int rime_fringe_ant_bwd_prepare(const float* gvis, int Nbl, int Nt, int Nf,
                                           void* workspace, size_t workspace_bytes, void* stream)
{
    if (!gvis || Nbl <= 0 || Nt <= 0 || Nf <= 0) return RIME_EINVAL;
    if (!workspace || workspace_bytes < rime_fringe_ant_bwd_workspace(Nbl, Nt, Nf)) return RIME_EWORKSPACE;
    hipLaunchKernelGGL(transpose_gvis_kernel, dim3((Nbl + 31) / 32, (Nf + 31) / 32, Nt), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), gvis, (float*)workspace, Nbl, Nt, Nf);
    return check_launch();
}

extern "C" int rime_fringe_ant_bwd_block(const double* antpos, int Nrows, int cross, int mirror, const double* sdir,
                                         const double* freqs, const float* gscale, const int* pair_direct,
                                         const int* pair_conj, int Nbl, int Nt, int Nf, int Pstride,
                                         long long st_t, long long st_f, long long st_p, int sign,
                                         int psky_complex, int accumulate, float* gpsky, const void* workspace,
                                         size_t workspace_bytes, void* stream)
{
    if (!antpos || !sdir || !freqs || !gscale || !pair_direct || !pair_conj || !gpsky) return RIME_EINVAL;
    if (!ant_common_ok(Nrows, cross, Nbl, Nt, Nf, Pstride, st_p, sign, psky_complex)) return RIME_EINVAL;
    if (!workspace || workspace_bytes < rime_fringe_ant_bwd_workspace(Nbl, Nt, Nf)) return RIME_EWORKSPACE;
    AntBwdArgs A{};
    A.antpos = antpos; A.sdir = sdir; A.freqs = freqs; A.gvis = nullptr; A.gscale = gscale;
    A.pair_direct = pair_direct; A.pair_conj = pair_conj; A.gpsky = gpsky; A.gvt = (const float*)workspace;
    A.Nant = Nrows; A.Nbl = Nbl; A.Nt = Nt; A.Nf = Nf; A.Pstride = Pstride;
    A.st_t = st_t; A.st_f = st_f; A.st_p = st_p; A.sign = (double)sign; A.accumulate = accumulate ? 1 : 0;
    A.rows_i = cross; A.imsign = psky_complex < 0 ? -1.f : 1.f;
    if (mirror < 0 || (Nrows > 0 && Nrows <= MF_NA && (mirror >> ((Nrows + 15) / 16)) != 0)) return RIME_EINVAL;
    A.mirror = cross ? 0 : mirror;                       // diagonal blocks, real or complex psky
    // pixel ranges are independent outputs: split freely for parallelism (>= 256 pixel tiles/block
    // amortise the G staging; fewer when the grid would otherwise be small)
    const int ntile = Pstride / 32;
    int per = 256;
    while (per > 8 && (long)Nt * Nf * ((ntile + per - 1) / per) < 1024) per /= 2;
    A.tiles_per_split = per;
    A.S = (ntile + per - 1) / per;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    dim3 grid((unsigned)Nt * A.S * Nf, 1, 1);
    if (cross) {
        if (psky_complex) hipLaunchKernelGGL(fringe_ant_bwd_cross_kernel<true>, grid, dim3(512), MX_LDS, st, A);
        else hipLaunchKernelGGL(fringe_ant_bwd_cross_kernel<false>, grid, dim3(512), MX_LDS, st, A);
    } else {
        const bool small = Nrows <= 64 && bwd_small_enabled();
        if (psky_complex) {
            if (small) hipLaunchKernelGGL((fringe_ant_bwd_kernel<true, 2>), grid, dim3(512), MB_LDS, st, A);
            else hipLaunchKernelGGL((fringe_ant_bwd_kernel<true, 4>), grid, dim3(512), MB_LDS, st, A);
        } else {
            if (small) hipLaunchKernelGGL((fringe_ant_bwd_kernel<false, 2>), grid, dim3(512), MB_LDS, st, A);
            else hipLaunchKernelGGL((fringe_ant_bwd_kernel<false, 4>), grid, dim3(512), MB_LDS, st, A);
        }
    }
    return check_launch();
}

extern "C" int rime_fringe_ant_bwd(const double* antpos, const double* sdir, const double* freqs,
                                   const float* gvis, const float* gscale, const int* pair_direct,
                                   const int* pair_conj, int Nant, int Nbl, int Nt, int Nf, int Pstride,
                                   long long st_t, long long st_f, long long st_p, int sign, float* gpsky,
                                   void* workspace, size_t workspace_bytes, void* stream)
{
    const int rc = rime_fringe_ant_bwd_prepare(gvis, Nbl, Nt, Nf, workspace, workspace_bytes, stream);
    if (rc != RIME_OK) return rc;
    return rime_fringe_ant_bwd_block(antpos, Nant, 0, 0, sdir, freqs, gscale, pair_direct, pair_conj, Nbl, Nt, Nf,
                                     Pstride, st_t, st_f, st_p, sign, 0, 0, gpsky, workspace, workspace_bytes, stream);
}

// ---- conjugate-pair form (fringe_pair_fwd_kernel / fringe_pair_bwd_kernel): one block of up to 64 rows -------------------
static bool pair_common_ok(int Nrows, int Nbl, int Nt, int Nf, int Pstride, long long st_p, int sign)
{
    return Nrows > 0 && Nrows <= 64 && ant_common_ok(Nrows, 0, Nbl, Nt, Nf, Pstride, st_p, sign, 0);     // st_p 1, or 2: one plane of a complex buffer
}

extern "C" int rime_fringe_pair_fwd_block(const double* antpos, int Nrows, const int* centre, int flat, const double* sdir,
                                          const double* freqs, const float* psky, const float* scale, const float* rowmin,
                                          const int* pair_direct, const int* pair_conj, int Nbl, int Nt, int Nf, int Pstride,
                                          long long st_t, long long st_f, long long st_p, int sign,
                                          void* workspace, size_t workspace_bytes, void* stream)
{
    if (!antpos || !sdir || !freqs || !psky || !scale || !pair_direct || !pair_conj) return RIME_EINVAL;
    if (!pair_common_ok(Nrows, Nbl, Nt, Nf, Pstride, st_p, sign)) return RIME_EINVAL;
    PairArgs A{};
    A.antpos = antpos; A.sdir = sdir; A.freqs = freqs; A.psky = psky; A.scale = scale; A.rowmin = rowmin;
    A.pair_direct = pair_direct; A.pair_conj = pair_conj; A.vis = nullptr; A.ws = (float*)workspace;
    A.Nant = Nrows; A.Nbl = Nbl; A.Nt = Nt; A.Nf = Nf; A.Pstride = Pstride;
    A.st_t = st_t; A.st_f = st_f; A.st_p = st_p; A.sign = (double)sign; A.imsign = 1.f; A.mirror = 0;
    A.centre = centre;
    ant_split_plan(Nt, Nf, Pstride, A.S, A.panels_per_split);
    if (!workspace || workspace_bytes < rime_fringe_ant_workspace(Nbl, Nt, Nf, Pstride)) return RIME_EWORKSPACE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    dim3 grid((unsigned)Nt * A.S * Nf, 1, 1);
    if (Nrows <= 32 && !centre) {                        // one row tile
        // LDS: two image buffers (18.5 KB) during the loop, then the epilogue's transposition tiles + two partial tiles per wave pair
        constexpr size_t lds1 = 4 * 33 * 32 * 4 + 2 * 2 * 16 * 64 * 4;
        static_assert(lds1 >= 2 * (size_t)FwdShape<1, 1, false>::BUF, "the image buffers fit into the epilogue's scratch");
        if (flat) {
            hipLaunchKernelGGL((fringe_pair_fwd1_kernel<true, true>), grid, dim3(256), lds1, st, A);
            if (rowmin) hipLaunchKernelGGL((fringe_pair_fwd1_kernel<false, true>), grid, dim3(256), lds1, st, A);
        } else {
            hipLaunchKernelGGL((fringe_pair_fwd1_kernel<true, false>), grid, dim3(256), lds1, st, A);
            if (rowmin) hipLaunchKernelGGL((fringe_pair_fwd1_kernel<false, false>), grid, dim3(256), lds1, st, A);
        }
        return check_launch();
    }
    using SH = FwdShape<2, 2, false>;
    static_assert(SH::LDS >= 4 * 33 * 32 * 4 + 2 * 64 * 2 * 4, "epilogue scratch: transposition tiles + the hub's column sums");
#define RIME_PAIR_FWD(CEN_, FLAT_)                                                                                       \
    do {                                                                                                                 \
        hipLaunchKernelGGL((fringe_pair_fwd_kernel<true, CEN_, FLAT_>), grid, dim3(256), SH::LDS, st, A);                \
        if (rowmin) hipLaunchKernelGGL((fringe_pair_fwd_kernel<false, CEN_, FLAT_>), grid, dim3(256), SH::LDS, st, A);   \
    } while (0)
    if (centre) { if (flat) RIME_PAIR_FWD(true, true); else RIME_PAIR_FWD(true, false); }
    else { if (flat) RIME_PAIR_FWD(false, true); else RIME_PAIR_FWD(false, false); }
#undef RIME_PAIR_FWD
    return check_launch();
}

extern "C" int rime_fringe_pair_bwd_block(const double* antpos, int Nrows, const int* centre, int flat, const double* sdir,
                                          const double* freqs, const float* gscale, const int* pair_direct,
                                          const int* pair_conj, int Nbl, int Nt, int Nf, int Pstride,
                                          long long st_t, long long st_f, long long st_p, int sign, int accumulate,
                                          float* gpsky, const void* workspace, size_t workspace_bytes, void* stream)
{
    if (!antpos || !sdir || !freqs || !gscale || !pair_direct || !pair_conj || !gpsky) return RIME_EINVAL;
    if (!pair_common_ok(Nrows, Nbl, Nt, Nf, Pstride, st_p, sign)) return RIME_EINVAL;
    if (!workspace || workspace_bytes < rime_fringe_ant_bwd_workspace(Nbl, Nt, Nf)) return RIME_EWORKSPACE;
    PairBwdArgs A{};
    A.antpos = antpos; A.sdir = sdir; A.freqs = freqs; A.gvis = nullptr; A.gscale = gscale;
    A.pair_direct = pair_direct; A.pair_conj = pair_conj; A.gpsky = gpsky; A.gvt = (const float*)workspace;
    A.Nant = Nrows; A.Nbl = Nbl; A.Nt = Nt; A.Nf = Nf; A.Pstride = Pstride;
    A.st_t = st_t; A.st_f = st_f; A.st_p = st_p; A.sign = (double)sign; A.accumulate = accumulate ? 1 : 0;
    A.rows_i = 0; A.imsign = 1.f; A.mirror = 0; A.centre = centre;
    const int ntile = Pstride / 32;
    int per = 256;
    while (per > 8 && (long)Nt * Nf * ((ntile + per - 1) / per) < 1024) per /= 2;
    A.tiles_per_split = per;
    A.S = (ntile + per - 1) / per;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    dim3 grid((unsigned)Nt * A.S * Nf, 1, 1);
    if (Nrows <= 32 && !centre) {                        // one row tile
        if (flat) hipLaunchKernelGGL((fringe_pair_bwd_kernel<false, true, 1>), grid, dim3(PB_THREADS), pb_lds<1>(), st, A);
        else hipLaunchKernelGGL((fringe_pair_bwd_kernel<false, false, 1>), grid, dim3(PB_THREADS), pb_lds<1>(), st, A);
    } else if (centre) {
        if (flat) hipLaunchKernelGGL((fringe_pair_bwd_kernel<true, true, 2>), grid, dim3(PB_THREADS), pb_lds<2>(), st, A);
        else hipLaunchKernelGGL((fringe_pair_bwd_kernel<true, false, 2>), grid, dim3(PB_THREADS), pb_lds<2>(), st, A);
    } else {
        if (flat) hipLaunchKernelGGL((fringe_pair_bwd_kernel<false, true, 2>), grid, dim3(PB_THREADS), pb_lds<2>(), st, A);
        else hipLaunchKernelGGL((fringe_pair_bwd_kernel<false, false, 2>), grid, dim3(PB_THREADS), pb_lds<2>(), st, A);
    }
    return check_launch();
}
