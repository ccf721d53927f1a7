/*
 * rime_hip.h -- C ABI of the MI355X-native RIME visibility-synthesis library (librime_hip.so)
 *
 * This is the drop-in boundary for the hot path of BayesLIM's `rime_model.RIME.forward`
 * and its autograd backward.  The reference has no FFI: its "interface" for this path is a
 * handful of Python tensor functions.  Each entry point below replaces the arithmetic of the
 * reference function named in its comment (file:line under /root/reference/bayeslim/).
 *
 * Conventions (all entry points):
 *   - extern "C", plain pointers and sizes; no torch / C++ types.
 *   - every pointer is a DEVICE pointer unless the parameter name ends in `_host`.
 *   - asynchronous on `stream` (a hipStream_t passed as void*); no allocation, no
 *     synchronisation, no host<->device copies: graph-capturable.  Workspace is caller-owned.
 *   - returns 0 on success, a negative RIME_E* code on a rejected call (never throws).
 *   - `dtype`: RIME_F32 (float / complex64) or RIME_F64 (double / complex128) for the
 *     psky / vis / map tensors.  Geometry (blvecs, sdir, freqs) is always float64.
 *   - tensors are dense, last index fastest, layouts given per function.
 */
#ifndef RIME_HIP_H
#define RIME_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RIME_F32 0
#define RIME_F64 1

#define RIME_OK          0
#define RIME_EINVAL     -1   /* bad shape / flag / null pointer            */
#define RIME_EWORKSPACE -2   /* workspace too small                         */
#define RIME_ELAUNCH    -3   /* hipLaunchKernel reported an error           */
#define RIME_EUNSUPPORTED -4

/* library / build identification: "rime_hip <version> gfx950" */
const char* rime_version(void);

/* last HIP error string seen by a failed launch (thread-unsafe diagnostic) */
const char* rime_last_error(void);

/* ---------------------------------------------------------------------------------------
 * Fringe sum, forward.
 *   vis[pp, b, t, f] = sum_p  psky[t, mp(b), pp, f, p] * exp(sign * 2 pi i * freq[f]/c * blvecs[b] . sdir[t, :, p])
 * Replaces: ArrayModel.gen_fringe (telescope_model.py:310-358) fused with the product and
 * pixel sum of RIME._prod_and_sum (rime_model.py:423-429).  The (Nbl, Nf, P) fringe tensor is
 * never materialised.
 *
 *   blvecs   f64 [Nbl, 3]            baseline vectors, ENU metres
 *   sdir     f64 [Nt, 3, Pstride]    unit pointing vectors per time step (x=E, y=N, z=Up);
 *                                    columns p >= npix of a time step must be finite (zero)
 *   freqs    f64 [Nf]                Hz
 *   psky     T   [Nt, Nmp, Npp, Nf, Pstride]      (real)   or
 *            T   [Nt, Nmp, Npp, Nf, Pstride, 2]   (complex, interleaved) perceived sky =
 *                                    apply_beam output per beam-model pair; padded columns 0.
 *            psky_strides_host: NULL for that dense layout, or 4 element strides (HOST)
 *                                    {time, model pair, pol product, channel} of a permuted /
 *                                    strided buffer; the pixel axis is always contiguous.
 *   mp_offsets_host  int[Nmp+1] (HOST) baselines of model pair g are bl_order[off[g] .. off[g+1])
 *   bl_order int [Nbl] or NULL       baseline index per slot (NULL: identity; needs Nmp == 1
 *                                    or baselines already grouped by model pair)
 *   vis      complex<T> [Npp, Nbl, Nt, Nf]  (interleaved re, im)
 *   freq_uniform_host: 0 = arbitrary channel centres (one sincos per element);
 *                    1 = exactly uniform grid freq0_host + k * dfreq_host (rotation recurrence);
 *                    2 = near-uniform: freqs[k] = freq0 + k * dfreq + eps_k with
 *                        2 pi max|eps_k| max_blen / c < 2e-3 (e.g. a float32-rounded linspace):
 *                        recurrence on the fitted grid + first-order per-channel correction.
 *   max_blen_host: upper bound on |blvecs[b]| [m] (<= 0: unknown).  Lets the library pick the
 *                    3-FMA shear rotation when max_blen * |dfreq| / c < 0.3 turn per channel.
 *   workspace: at least rime_fringe_sum_workspace(...) bytes.
 * ------------------------------------------------------------------------------------- */
size_t rime_fringe_sum_workspace(int dtype, int Nbl, int Nt, int Nf, int Pstride,
                                 int Nmp, int Npp, int psky_complex, int backward);

int rime_fringe_sum_fwd(int dtype,
                        const double* blvecs, const double* sdir, const double* freqs,
                        const void* psky, const int* mp_offsets_host, const int* bl_order,
                        int Nbl, int Nt, int Nf, int Pstride, int Nmp, int Npp,
                        int psky_complex, int sign,
                        int freq_uniform_host, double freq0_host, double dfreq_host,
                        double max_blen_host, const long long* psky_strides_host,
                        void* vis, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------
 * Fringe sum, backward (gradient w.r.t. psky; PyTorch convention grad = dL/dRe + i dL/dIm):
 *   gpsky[t, mp, pp, f, p] = sum_{b in group mp} conj(F[b,t,f,p]) * gvis[pp, b, t, f]
 *   (real part only when psky is real).  Regenerates the fringe; deterministic (no atomics).
 * Replaces the autograd backward of rime_model.py:429 (SumBackward/MulBackward over the
 * saved (Nbl,Nf,P) fringe and psky tensors).
 *   gvis  complex<T> [Npp, Nbl, Nt, Nf];  gpsky same layout as psky (fully overwritten).
 * ------------------------------------------------------------------------------------- */
int rime_fringe_sum_bwd(int dtype,
                        const double* blvecs, const double* sdir, const double* freqs,
                        const void* gvis, const int* mp_offsets_host, const int* bl_order,
                        int Nbl, int Nt, int Nf, int Pstride, int Nmp, int Npp,
                        int psky_complex, int sign,
                        int freq_uniform_host, double freq0_host, double dfreq_host,
                        double max_blen_host, const long long* gpsky_strides_host,
                        void* gpsky, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------
 * Antenna-factored fringe sum on the matrix cores (float32, one beam-model pair).  Same result as rime_fringe_sum_fwd/bwd for baselines that are antenna
 * pairs: with E_a = exp(sign 2 pi i nu r_a.s / c) the fringe of baseline (a1 -> a2) is
 * E_a2 conj(E_a1), so per (time, channel) all visibilities are one Hermitian rank-P update
 * V = E^H diag(psky) E, computed on v_mfma_f32_32x32x16_f16 with an f16 hi/lo split of the f32
 * operands (three cross products, f32 accumulation).  Replaces the same reference lines as
 * rime_fringe_sum_fwd/bwd.
 *   antpos f64 [Nant, 3] ENU metres; psky / gpsky f32 [t][f][p] with element strides st_t, st_f,
 *       st_p (1, or 2 for the real / imaginary plane of an interleaved complex buffer).  One call
 *       handles one real plane: polarisation products and the two planes of a complex psky are
 *       separate calls (V is linear in psky: V[ar + i ai] = V[ar] + i V[ai])
 *   scale / gscale f32 [Nt, Nf]: power-of-two factors that bring max|psky[t,f,:]| (resp.
 *       max|gvis[:,t,f]|) to ~2^14 -- computed by the caller (a torch amax), exact to undo
 *   rowmin f32 [Nt, Nf] or NULL (forward): min of each psky row; rows with min >= 0 run without the
 *       per-pixel sign masks (NULL: every row is treated as signed)
 *   pair_direct / pair_conj int32 [128*128]: for antenna indices (i, j) with tile(i) <= tile(j)
 *       (tile = index / 32) the baseline slot that receives V[i,j] (pair stored as i -> j) and the
 *       slot that receives conj(V[i,j]) (pair stored as j -> i, only when tile(j) > tile(i)), or -1
 *   vis / gvis complex64 [Nbl, Nt, Nf]
 * ------------------------------------------------------------------------------------- */
size_t rime_fringe_ant_workspace(int Nbl, int Nt, int Nf, int Pstride);   /* forward: per-split result slabs */
size_t rime_fringe_ant_bwd_workspace(int Nbl, int Nt, int Nf);              /* backward: transposed gvis */
int rime_fringe_ant_fwd(const double* antpos, const double* sdir, const double* freqs,
                        const float* psky, const float* scale, const float* rowmin,
                        const int* pair_direct,
                        const int* pair_conj, int Nant, int Nbl, int Nt, int Nf, int Pstride,
                        long long st_t, long long st_f, long long st_p, int sign, float* vis,
                        void* workspace, size_t workspace_bytes, void* stream);
int rime_fringe_ant_bwd(const double* antpos, const double* sdir, const double* freqs,
                        const float* gvis, const float* gscale, const int* pair_direct,
                        const int* pair_conj, int Nant, int Nbl, int Nt, int Nf, int Pstride,
                        long long st_t, long long st_f, long long st_p, int sign, float* gpsky,
                        void* workspace, size_t workspace_bytes, void* stream);

/* Block decomposition of the pair matrix.  The antennas are cut into groups (<= 128 antennas: by
 * position for arrays with more than 128 antennas, by beam model when antennas carry different beam
 * models -- beam_model.py:303-327 pairs them per baseline --, by rank-local tile for baseline-tile
 * sharding across GPUs) and the pair matrix into blocks; every block is one launch with its own psky
 * plane.  rime_fringe_ant_fwd == one diagonal block + finish; rime_fringe_ant_bwd == prepare + one
 * diagonal block.
 *   diagonal block (cross = 0): antpos [Nrows <= 128, 3] of one group against itself, tables as above
 *       (local indices)
 *   cross block (cross = rows of group I): antpos [Nrows, 3] = group I (cross rows) then group J
 *       (Nrows - cross rows), each zero-padded to a multiple of 32; supported (rows I, rows J): (32, 32),
 *       (32, 64), (64, 64), (128, 128); pair_direct[i*128 + j] = slot of baseline (I_i -> J_j),
 *       pair_conj[i*128 + j] = slot of baseline (J_j -> I_i), or -1
 *   self block (forward only, cross == Nrows in {32, 64, 96, 128}, psky_complex != 0): the diagonal block of one
 *       group as the cross block of the group with ITSELF -- antpos [Nrows, 3] zero-padded, the tables of the
 *       diagonal block (entries i < j); L and B images of the same antennas come from one evaluation of the
 *       phase, only the upper-triangular tiles are contracted
 *   psky_complex: 0 = psky / gpsky is one real plane (st_p 1 or 2); +1 = interleaved complex (st_p == 2)
 *       handled in ONE pass (forward: cross and self blocks -- a diagonal block (cross = 0) returns
 *       RIME_EUNSUPPORTED and takes one call per real plane; the block must hold direct entries only); -1 = as +1 but the
 *       block contracts conj(psky) (a block built with its groups swapped: conj entries only).  The
 *       backward writes both gradient planes from one pass for either block kind.
 *   mirror: bit mask over the 16-row groups g of a DIAGONAL block (0 = none; round 5).  Bit g set: rows 16 g + 8 + i hold the
 *       MIRROR antennas of rows 16 g + i, i < 8 -- antpos[16 g + 8 + i] = -antpos[16 g + i] (the block's positions are
 *       measured from the centre of symmetry; visibilities depend on position differences only), rows without an antenna
 *       in either octet zero.  Their phasors are complex conjugates: the kernels evaluate the first octet and conjugate
 *       it for the second (forward: the real-plane passes of every diagonal block -- the 33..48-row shape in its first row
 *       tile only; backward: every diagonal block).  A licence, not an obligation: other block kinds evaluate every row.  Bits at or beyond ceil(Nrows / 16) -> RIME_EINVAL.
 * Forward blocks fill disjoint baseline slots of the slab workspace (every baseline must belong
 * to exactly one block); _finish sums the pixel splits and writes vis [Nbl, Nt, Nf].  Backward:
 * _prepare transposes gvis into the workspace once, every block reads it; blocks after the first that
 * write the same gpsky plane pass accumulate = 1 (stream order makes the sum deterministic). */
int rime_fringe_ant_fwd_block(const double* antpos, int Nrows, int cross, int mirror, const double* sdir,
                              const double* freqs, const float* psky, const float* scale,
                              const float* rowmin, const int* pair_direct, const int* pair_conj, int Nbl, int Nt, int Nf,
                              int Pstride, long long st_t, long long st_f, long long st_p, int sign,
                              int psky_complex, void* workspace, size_t workspace_bytes, void* stream);
/* Row pre-scales of the matrix-core kernels in one launch: for every row (i0, i1, i2, i3) of a 4-D arrangement of
 * contiguous float32 rows of length L at x + i0 s0 + i1 s1 + i2 s2 + i3 s3 (element strides):
 *   scale[row]  = 2^floor(log2(2^14 / max|x|)) (1 for an all-zero row)   -- the `scale` / `gscale` inputs above
 *   rowmin[row] = min x (or NULL)                                          -- the `rowmin` input of the forward
 * rows numbered ((i0 d1 + i1) d2 + i2) d3 + i3.  Host-side convenience (what the shipped binding computes these two
 * inputs with); the reference has no counterpart. */
int rime_fringe_row_scale(const float* x, int d0, int d1, int d2, int d3, long long s0, long long s1,
                          long long s2, long long s3, int L, float* scale, float* rowmin, void* stream);
/* The same for rows of an INTERLEAVED COMPLEX float32 psky (full-polarisation layouts; strides and L in complex elements):
 * scale[row] from max(|re|, |im|) over the row, rowmin_re / rowmin_im (or NULL) the minimum of each plane -- the `rowmin`
 * inputs of the per-plane forward passes. */
int rime_fringe_row_scale_cplx(const float* x, int d0, int d1, int d2, int d3, long long s0, long long s1,
                               long long s2, long long s3, int L, float* scale, float* rowmin_re, float* rowmin_im,
                               void* stream);
int rime_fringe_ant_fwd_finish(const void* workspace, size_t workspace_bytes, float* vis,
                               int Nbl, int Nt, int Nf, int Pstride, void* stream);
int rime_fringe_ant_bwd_prepare(const float* gvis, int Nbl, int Nt, int Nf,
                                void* workspace, size_t workspace_bytes, void* stream);
int rime_fringe_ant_bwd_block(const double* antpos, int Nrows, int cross, int mirror, const double* sdir,
                              const double* freqs, const float* gscale, const int* pair_direct,
                              const int* pair_conj, int Nbl, int Nt, int Nf, int Pstride,
                              long long st_t, long long st_f, long long st_p, int sign,
                              int psky_complex, int accumulate, float* gpsky, const void* workspace,
                              size_t workspace_bytes, void* stream);

/* Conjugate-pair form of a diagonal block, real psky (round 5; csrc/fringe_mfma.hip, "CONJUGATE-PAIR FORM").  Replaces the same
 * reference lines as the blocks above (telescope_model.py:310-358 + rime_model.py:423-429) for an array with POINT SYMMETRY:
 * antennas that come in mirror pairs about a centre c, r' - c = -(r - c), have conjugate phasors, and all pairs of up to
 * 128 (+ 1) antennas follow from the phasors of one antenna of each pair.
 *   antpos [Nrows <= 64, 3]: positions MEASURED FROM c of the "firsts" -- one antenna of every mirror pair -- and of the
 *       antennas without a partner, one per row.
 *   pair_direct / pair_conj: the tables of a VIRTUAL diagonal block of 128 rows -- row i < 64 = the antenna of antpos row i,
 *       row 64 + i = its mirror antenna (no antenna: no entries) -- built by the rule of every diagonal block
 *       (pair_direct[i*128 + j] = slot of baseline i -> j for tile(i) <= tile(j), else pair_conj[j*128 + i]).
 *   centre: NULL, or int32 [2][128] for ONE more antenna that sits at c itself (its phasor is 1; needed when the firsts and
 *       singles already fill 64 rows): centre[r] = slot of the baseline (hub -> virtual row r), centre[128 + r] = slot of
 *       (virtual row r -> hub), or -1.
 *   flat: nonzero = every antpos z is zero (a coplanar array measured from a centre in its plane): the kernels do not evaluate that
 *       term of the phase.  A licence stated by the caller; 0 is always correct.
 *   psky / gpsky is ONE real plane (st_p = 1, or 2 for a plane of an interleaved complex buffer: V is linear in psky, a complex
 *   psky takes one call per plane as on every diagonal block); the other arguments, the workspace and _finish / _prepare are those of rime_fringe_ant_fwd_block /
 *   rime_fringe_ant_bwd_block, and pair blocks mix with other blocks of the same launch sequence. */
int rime_fringe_pair_fwd_block(const double* antpos, int Nrows, const int* centre, int flat, const double* sdir,
                               const double* freqs, const float* psky, const float* scale, const float* rowmin,
                               const int* pair_direct, const int* pair_conj, int Nbl, int Nt, int Nf, int Pstride,
                               long long st_t, long long st_f, long long st_p, int sign,
                               void* workspace, size_t workspace_bytes, void* stream);
int rime_fringe_pair_bwd_block(const double* antpos, int Nrows, const int* centre, int flat, const double* sdir,
                               const double* freqs, const float* gscale, const int* pair_direct,
                               const int* pair_conj, int Nbl, int Nt, int Nf, int Pstride,
                               long long st_t, long long st_f, long long st_p, int sign, int accumulate,
                               float* gpsky, const void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------
 * ICRS (ra, dec) -> topocentric (zenith angle, azimuth East of North), degrees, float64.
 * Replaces the per-direction part of telescope_model.eq2top (telescope_model.py:469-502,
 * astropy ICRS -> AltAz): annual aberration, rotation by the caller's 3 x 3 matrix
 * M = L(lat) R3(GAST + lon) N P B (ICRS -> East, North, Up; bayeslim_amd/astrometry.py builds it
 * per observation time: IAU 2006 precession + frame bias, truncated IAU 1980 nutation, GAST),
 * diurnal aberration, conversion to angles.
 *   ra_deg, dec_deg, zen_deg, az_deg: device f64 [N];  M_host f64 [9] row-major and
 *   vbary_host f64 [3] (observer velocity / c, ICRS axes) are HOST pointers; vdiurnal = eastward
 *   site velocity / c
 * ------------------------------------------------------------------------------------- */
int rime_eq2top(const double* ra_deg, const double* dec_deg, int N, const double* M_host,
                const double* vbary_host, double vdiurnal, double* zen_deg, double* az_deg, void* stream);

/* ---------------------------------------------------------------------------------------
 * Materialised fringe, for callers that want the tensor itself (imaging.VisMapper.build_A,
 * tests):  out[b, f, p] = exp(sign * 2 pi i * freqs[f]/c * blvecs[b] . sdir[:, p])
 * Replaces ArrayModel.gen_fringe (telescope_model.py:350-356).  RIME never calls it.
 *   sdir f64 [3, sdir_stride];  out complex<T> [Nbl, Nf, P]
 * ------------------------------------------------------------------------------------- */
int rime_gen_fringe(int dtype, const double* blvecs, const double* sdir, const double* freqs,
                    int Nbl, int Nf, int P, int sdir_stride, int sign, void* out, void* stream);

/* ---------------------------------------------------------------------------------------
 * Pixel-beam interpolation gather:  out[r, p] = sum_k wgts[p, k] * m[r, inds[p, k]]
 * Replaces PixInterp.interp (utils.py:815-861: index_select + einsum).
 *   m    T [R, Npb]   (R = product of leading dims: Npol*Nvec*Nmodel*Nf), or complex<T>
 *   inds int32 [P, Nnn];  wgts T [P, Nnn];  out T [R, out_stride] (columns >= P untouched)
 * ------------------------------------------------------------------------------------- */
int rime_interp_gather_fwd(int dtype, int is_complex, const void* m, const int* inds,
                           const void* wgts, int R, int Npb, int P, int Nnn,
                           void* out, int out_stride, void* stream);

/* Adjoint of the gather, deterministic: gm[r, j] = sum over (p,k) with inds[p,k]==j of
 * wgts[p,k] * gout[r, p], driven by a CSR inverse index built once per (inds, wgts):
 *   csr_ptr int32 [Npb+1], csr_src int32 [nnz] (flat p*Nnn+k positions, ascending per row).
 * Both tensors are passed TRANSPOSED so that every access is coalesced:
 *   goutT T [P, R] (complex: [P, R, 2]) -- all map rows of one sky pixel contiguous
 *   gmT   T [Npb, R] (complex: [Npb, R, 2])
 * Replaces the index_select/einsum backward (scatter-add) of utils.py:833-841. */
int rime_interp_scatter_bwd(int dtype, int is_complex, const void* goutT,
                            const int* csr_ptr, const int* csr_src, const void* wgts,
                            int R, int Npb, int P, int Nnn, void* gmT, void* stream);
/* The same adjoint for a ONE-NODE stencil (Nnn = 1: the FoV cut, cut_sky_fov beam_model.py:1681-1698, and the redundant
 * inflation, rime_model.py:436-437) on ROW-MAJOR buffers -- no transposed copies:
 *   gout T [R, gout_stride] (complex: [R, gout_stride, 2]), gm T [R, Npb]; csr_src holds point indices q, wgts T [P].
 *   Any R (rows beyond one grid's reach take further launches); an EMPTY index (csr_ptr all zero: every weight 0) may pass
 *   csr_src = NULL and yields gm = 0. */
int rime_interp_scatter_rows_bwd(int dtype, int is_complex, const void* gout, long long gout_stride,
                                 const int* csr_ptr, const int* csr_src, const void* wgts, int R, int Npb,
                                 void* gm, void* stream);

/* ---------------------------------------------------------------------------------------
 * Fused psky builder (1-pol power beam, one beam model):
 *   psky[r, q] = ( sum_k wgts[q,k] * bmapT[inds[q,k], r] ) * sky[r, cut[q]],  r = channel, q = (t, p)
 * = PixelBeam.gen_beam's interpolation (beam_model.py:238-269) + cut_sky_fov (:1681-1698) + the
 * beam x sky product of apply_beam (:313-322) in one pass; cut[q] == Npix marks zero padding.
 *   bmapT T [Npb, R] (the beam map NODE-MAJOR: a node's channels are contiguous, so the gathers are vector
 *   loads); sky T [R, Npix]; inds i32 / wgts T [Q = Nt*Ps, Nnn]; cut i32 [Q]; psky T [R, Q]
 * Backward: T1 [Q, R] = gpsky * sky_cut (transposed: the input layout of rime_interp_scatter_bwd, which
 * then yields the beam-map gradient); gsky [R, Npix] = sum over time steps of gpsky * (re-interpolated beam)
 * through pos i32 [Nt, Npix] (index of sky pixel j inside time step t's cut, or -1).  No atomics.
 * ------------------------------------------------------------------------------------- */
int rime_beam_sky_fwd(int dtype, const void* bmapT, const void* sky, const int* inds, const void* wgts,
                      const int* cut, int R, int Npb, int Npix, int Q, int Nnn, void* psky, void* stream);
/* workspace: partial planes of the sky gradient when its time steps are split over blocks (workloads of many time steps and
 * few sky pixels; 0 bytes when one block walks all steps).  NULL = no split. */
size_t rime_beam_sky_bwd_workspace(int dtype, int R, int Npix, int Nt);
int rime_beam_sky_bwd(int dtype, const void* gpsky, const void* bmapT, const void* sky, const int* inds,
                      const void* wgts, const int* cut, const int* pos, int R, int Npb, int Npix,
                      int Nt, int Ps, int Nnn, void* T1, void* gsky, void* workspace, size_t workspace_bytes,
                      void* stream);

/* ---------------------------------------------------------------------------------------
 * Full-polarisation beam x sky product, elementwise:  psky[a, d] = sum_{b, c} J1[a, b] S[b, c] conj(J2[d, c])
 * Replaces the 4-pol branch of PixelBeam.apply_beam (beam_model.py:345-363, einsum "ab...,bc...,dc...->ad...") and
 * its autograd backward in one pass each (the torch composition materialises two temporaries of 8 x psky).
 *   J1, J2 : T [2, 2, N] (beam_complex = 0) or complex<T> [2, 2, N];  J2 may be the same pointer as J1
 *   S      : complex<T> [2, 2, Ns], Ns dividing N (one sky for all Nmp model pairs: element n reads n % Ns)
 *   out, G : complex<T> [2, 2, N]  (N = Nmp * Nf * P)
 * Backward: gS [2, 2, N] = J1^H G J2 (the caller sums over model pairs when Ns < N), gJ1 = G (S J2^H)^H,
 * gJ2 = G^H (J1 S) -- real parts for a real beam -- each [2, 2, N] in the beam's type.  The caller adds gJ1 + gJ2 when
 * J2 is J1.
 * ------------------------------------------------------------------------------------- */
int rime_jones_apply_fwd(int dtype, int beam_complex, const void* J1, const void* J2, const void* S,
                         long long N, long long Ns, void* out, void* stream);
int rime_jones_apply_bwd(int dtype, int beam_complex, const void* J1, const void* J2, const void* S,
                         const void* G, long long N, long long Ns, void* gJ1, void* gJ2, void* gS, void* stream);

/* Stokes I + fractional polarisation -> coherency matrix, one pass each way
 *   C = I [[1 + fQ, fU - i fV], [fU + i fV, 1 - fQ]]
 * (sky_model.py:1160-1300: Stokes2Coherency on a Stokes-I sky with fractions; the reference composes it from ~15 tensor ops).
 *   stokesI : T [R, P] contiguous;  frac : T, element f_k[r, p] at frac[k fs_k + r fs_r + p fs_p] (strides 0 = broadcast), k = Q, U, V
 *   coh, gcoh : complex<T> [2, 2, R, P];   gI : T [R, P] = (1 + fQ) Re g00 + (1 - fQ) Re g11 + fU (Re g01 + Re g10) + fV (Im g10 - Im g01) */
int rime_stokes2coh_fwd(int dtype, const void* stokesI, const void* frac, long long fs_k, long long fs_r, long long fs_p,
                        long long R, long long P, void* coh, void* stream);
int rime_stokes2coh_bwd(int dtype, const void* gcoh, const void* frac, long long fs_k, long long fs_r, long long fs_p,
                        long long R, long long P, void* gI, void* stream);

/* ---------------------------------------------------------------------------------------
 * a_lm -> pixel transform:  out[r, j] = sum_c ( are[r,c] * Yre[c,j] - aim[r,c] * Yim[c,j] )
 *   = Re( (a * alm_mult) @ Ylm ), with alm_mult already folded into `alm` by the caller
 *   (an elementwise torch op, kept in autograd).
 * Replaces AlmModel.forward_alm (sph_harm.py:1342-1372), real_output=True branch.
 *   alm  T [R, Ncoeff, 2] (interleaved complex);  Ylm T [Ncoeff, Npix, 2];  out T [R, Npix]
 * Backward: galm[r, c] = sum_j gout[r, j] * conj(Ylm[c, j])   (complex, interleaved); the pixel
 *   axis may be split over blocks, partial sums go to a caller-owned workspace
 *   (rime_alm2pix_bwd_workspace bytes) and are reduced deterministically.
 * float32, y_scale > 0: operands split into f16 hi + lo halves on v_mfma_f32_32x32x16_f16 (three
 *   cross products, f32 accumulation, 22 significant bits) so both directions run at the rate Ylm
 *   streams from HBM.  y_scale is a power of two that brings max|Ylm| into [2^10, 2^14] (1.0 for
 *   orthonormal Ylm); the row operand is scaled per row inside the library.
 * float32, y_scale == 0: exact-f32 matrix cores (v_mfma_f32_32x32x2_f32).  float64: VALU.
 * Both directions take a caller-owned workspace (rime_alm2pix_*_workspace bytes).
 * ------------------------------------------------------------------------------------- */
size_t rime_alm2pix_fwd_workspace(int dtype, int R, int Ncoeff, int Npix);
int rime_alm2pix_fwd(int dtype, const void* alm, const void* Ylm, double y_scale, int R, int Ncoeff,
                     int Npix, void* out, void* workspace, size_t workspace_bytes, void* stream);
size_t rime_alm2pix_bwd_workspace(int dtype, int R, int Ncoeff, int Npix);
int rime_alm2pix_bwd(int dtype, const void* gout, const void* Ylm, double y_scale, int R, int Ncoeff,
                     int Npix, void* galm, void* workspace, size_t workspace_bytes, void* stream);

/* Packed Ylm (float32 f16-split path only): Ylm x y_scale split ONCE into f16 hi / lo halves, stored in the order the
 * matrix-core fragments are consumed -- 8 bytes per (coefficient, pixel), as the complex64 matrix itself, one copy per
 * direction (0 = forward: contraction over coefficients; 1 = backward: contraction over pixels).  The transforms on a
 * packed copy do no arithmetic on the streamed operand and read it as fully coalesced 1-KB fragments; the products are
 * those of rime_alm2pix_fwd / _bwd with y_scale > 0 (same split, same MFMAs): the forward is bitwise equal (same summation
 * order over K) as long as it runs WITHOUT a K split -- maps that offer fewer 4-wave blocks than 1.5 resident grids (the C3
 * shape: 2 splits) sum two partial planes in a second kernel, another order of the float32 sums (~2e-6 of the maximum) --, the backward
 * differs in the order of its float32 partial sums over pixels (~1e-6).  The caller owns the packed buffer (rime_alm2pix_packed_bytes) and must pack
 * again when Ylm or y_scale changes.  Workspaces: rime_alm2pix_fwd_workspace / rime_alm2pix_bwd_workspace bytes.
 * Same reference lines as above (sph_harm.py:1342-1372); the reference recomputes the einsum on the complex matrix. */
size_t rime_alm2pix_packed_bytes(int Ncoeff, int Npix, int direction);
int rime_alm2pix_pack(const void* Ylm, double y_scale, int Ncoeff, int Npix, int direction, void* packed, void* stream);
int rime_alm2pix_fwd_packed(const void* alm, const void* packed, double y_scale, int R, int Ncoeff, int Npix,
                            void* out, void* workspace, size_t workspace_bytes, void* stream);
int rime_alm2pix_bwd_packed(const void* gout, const void* packed, double y_scale, int R, int Ncoeff, int Npix,
                            void* galm, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------
 * Likelihood epilogue:  chi^2 = sum_i icov[i] * |pred[i] - data[i]|^2  over a complex visibility tensor
 * (N complex elements, interleaved), and its backward gpred[i] = 2 g icov[i] (pred[i] - data[i]).
 * Replaces `res = prediction - data; apply_icov(res, icov, cov_axis=None); torch.sum(...)` of
 * LogProb.forward_chisq (optim.py:1019-1027, apply_icov :1889-1894) -- SURVEY section 8(f) item 4.
 *   data NULL -> 0; icov T [N] real or NULL -> 1; out T [1]; g T [1] (device scalar: upstream gradient)
 * Deterministic two-pass reduction (double partial sums in a caller-owned workspace).
 * ------------------------------------------------------------------------------------- */
size_t rime_chisq_workspace(void);
int rime_chisq_fwd(int dtype, const void* pred, const void* data, const void* icov, size_t N,
                   void* out, void* workspace, size_t workspace_bytes, void* stream);
int rime_chisq_bwd(int dtype, const void* pred, const void* data, const void* icov, const void* g,
                   size_t N, void* gpred, void* stream);

/* ---------------------------------------------------------------------------------------
 * Gain application  V' = G1 V G2^dagger  per (baseline, time, channel) -- the post-RIME calibration
 * step, calibration._apply_cal (calibration.py:2412-2487; complex visibilities, no undo / covariance),
 * SURVEY section 8(f) item 3.  NP = 1 (1-pol) or 2; diag != 0 with NP = 2 is the reference's '2pol' mode
 * (diagonal products only, off-diagonal results zero: linalg.diag_matmul, linalg.py:116-149).
 *   vis, out, gout, gvis, d1, d2   T [NP][NP][Nbl][Nt][Nf][2]   contiguous, interleaved complex
 *   gains                          complex T, element (p, q, a, t, f) at complex offset
 *                                  (p*NP+q)*gst_p + a*gst_a + t*gst_t + f*gst_f  (0 strides broadcast)
 *   a1, a2                         int32 [Nbl]   antenna slots of each baseline (< Nant)
 * Backward (torch's conjugate-gradient convention): gvis = G1^dagger gout G2 and the PER-BASELINE gain
 * gradients d1 = gout (V G2^dagger)^dagger (belongs to antenna a1), d2 = gout^dagger (G1 V) (to a2); the
 * caller reduces d1 / d2 over the baselines of each antenna (and over broadcast axes).
 * ------------------------------------------------------------------------------------- */
int rime_apply_cal_fwd(int dtype, int NP, int diag, const void* vis, const void* gains, const int* a1,
                       const int* a2, int Nbl, int Nt, int Nf, int Nant, long long gst_p, long long gst_a,
                       long long gst_t, long long gst_f, void* out, void* stream);
int rime_apply_cal_bwd(int dtype, int NP, int diag, const void* vis, const void* gains, const void* gout,
                       const int* a1, const int* a2, int Nbl, int Nt, int Nf, int Nant, long long gst_p,
                       long long gst_a, long long gst_t, long long gst_f, void* gvis, void* d1, void* d2,
                       void* stream);

/* ---------------------------------------------------------------------------------------
 * Collectives of the sharded RIME step over RCCL (the replacement of DistributedLogProb.closure's per-device
 * Python loop, optim.py:1539-1566).  Thin wrappers: raw device pointers, the caller's stream, no allocation.
 * RCCL is resolved at first use (the copy already loaded into the process wins); without it every call
 * returns RIME_EUNSUPPORTED.  The shipped Python path drives the same RCCL operations through
 * torch.distributed; these give a host without torch.distributed the same two operations.
 *   rime_comm_unique_id : fills 128 bytes (ncclUniqueId) on ONE rank; the caller distributes them
 *   rime_comm_init      : ncclCommInitRank on the current device -> opaque communicator
 *   rime_comm_allgather_vis : every rank contributes `complex_per_rank` complex values (its block of the
 *       visibility tensor, blocks of equal size); vis_all receives nranks blocks in rank order
 *   rime_comm_reduce_grads  : in-place sum over ranks of `count` real values (gradients of replicated parameters)
 * ------------------------------------------------------------------------------------- */
int rime_comm_unique_id(void* id128);
int rime_comm_init(void** comm_out, int nranks, int rank, const void* id128);
int rime_comm_destroy(void* comm);
int rime_comm_allgather_vis(void* comm, int dtype, const void* vis_local, void* vis_all,
                            size_t complex_per_rank, void* stream);
int rime_comm_reduce_grads(void* comm, int dtype, void* grads, size_t count, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RIME_HIP_H */
