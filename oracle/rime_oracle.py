"""
CPU oracle for the RIME visibility-synthesis hot path.

*** TEST INFRASTRUCTURE -- NOT PART OF THE PRODUCT ***
Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this
module, and only as the checker / the timed CPU baseline.  The product (bayeslim_amd/)
never imports it and has no CPU fallback.

What it is: a plain PyTorch-CPU restatement, op for op, of the reference's algorithm for
the path `rime_model.RIME.forward` (+ autograd backward).  Each function cites the
reference lines it follows (paths relative to /root/reference/bayeslim/).  Gradients come
from torch autograd over these ops, exactly as in the reference (which has no custom
backward).  Run in float64 it is the parity oracle; run in the reference's default dtype
with all host threads it is the CPU baseline ("port").

Parity pinning: every function below is checked against golden vectors produced by the
imported reference itself (tests/golden/*.npz, generator tests/golden/make_golden.py):
tests/test_oracle_golden.py.  Two third-party pieces are NOT pinned because the packages are
absent from the reference checkout and from this image (astropy ICRS->AltAz; healpy
get_interp_weights) -- see DESIGN.md "parity unpinned".
"""
import math
import numpy as np
import torch

D2R = math.pi / 180.0
C_LIGHT = 2.99792458e8          # telescope_model.py:355 uses this literal


# ---------------------------------------------------------------------------
# a3: ArrayModel.gen_fringe                                telescope_model.py:310-358
# ---------------------------------------------------------------------------
def pointing_vectors(zen, az):
    """s = (sin z sin a, sin z cos a, cos z), az East of North; (3, P).  telescope_model.py:337-343"""
    z = zen * D2R
    a = az * D2R
    return torch.stack([torch.sin(z) * torch.sin(a), torch.sin(z) * torch.cos(a), torch.cos(z)])


def gen_fringe(blvecs, zen, az, freqs, conj=False):
    """exp(+-2 pi i nu/c b.s) -> (Nbl, Nf, P).  telescope_model.py:351-356"""
    s = pointing_vectors(zen, az).to(blvecs.dtype)
    sign = -2j if conj else 2j
    const = freqs[:, None] * (sign * math.pi / C_LIGHT)
    return ((blvecs @ s)[:, None, :] * const).exp_()          # in place on the temporary, as the reference (:356)


# ---------------------------------------------------------------------------
# a7: PixelBeam.apply_beam                                 beam_model.py:273-372
# ---------------------------------------------------------------------------
def apply_beam(beam, sky, bl_models, powerbeam):
    """
    beam (Npol, Nvec, Nmodel, Nf, P); sky (Nv, Nv, Nf, P); bl_models: list of
    (model1, model2) per baseline (the reference derives it from ant2beam, :303).
    Returns psky (Npol, Npol|1, Nbl, Nf, P).
    """
    Npol, Nvec = beam.shape[:2]
    pairs = sorted(set(bl_models))                                   # :304
    i1 = torch.as_tensor([p[0] for p in pairs])
    i2 = torch.as_tensor([p[1] for p in pairs])
    b1 = beam.index_select(2, i1)                                    # :314-327
    b2 = beam.index_select(2, i2)
    sk = sky[:, :, None]                                             # :330-331
    if Npol == 1 and Nvec == 1:
        assert tuple(sky.shape[:2]) == (1, 1)
        psky = b1 * sk if powerbeam else (b1 * b2.conj()) * sk       # :340-343
    elif powerbeam:
        assert Npol == 2 and Nvec == 1 and tuple(sky.shape[:2]) == (1, 1)
        psky = torch.stack([b1[0, 0] * sk[0, 0], b1[1, 0] * sk[0, 0]])[:, None]   # :355-357
    else:
        assert tuple(sky.shape[:2]) == (2, 2)
        # J_p B J_q^dagger: out[a,d] = sum_bc b1[a,b] sky[b,c] conj(b2[d,c])       # :347 / :363
        psky = torch.einsum('abmfp,bcmfp,dcmfp->admfp', b1.to(_ctype(b1, sk)),
                            sk.expand(-1, -1, len(pairs), -1, -1).to(_ctype(b1, sk)),
                            b2.conj().to(_ctype(b1, sk)))
    mp_idx = torch.as_tensor([pairs.index(p) for p in bl_models])    # :366-370
    return psky.index_select(2, mp_idx)


def _ctype(a, b):
    return torch.promote_types(a.dtype, b.dtype)


# ---------------------------------------------------------------------------
# a2: RIME._prod_and_sum                                   rime_model.py:391-440
# ---------------------------------------------------------------------------
def prod_and_sum(psky, blvecs, zen, az, freqs, sim2data_idx=None):
    """sum_pix fringe * psky -> (Npol, Npol|1, Nbl[_data], Nf).  rime_model.py:426-437"""
    fringe = gen_fringe(blvecs, zen, az, freqs)
    out = torch.sum(fringe * psky, dim=-1)
    if sim2data_idx is not None:
        out = out.index_select(2, sim2data_idx)
    return out


# ---------------------------------------------------------------------------
# a4/a8: FoV cut                                            beam_model.py:221-227, 1681-1698
# ---------------------------------------------------------------------------
def fov_cut(zen, fov=180.0):
    """indices with zen < fov/2 (strict), or all of them when fov >= 360.  beam_model.py:221-224"""
    if fov < 360:
        return torch.where(zen < fov / 2)[0]
    return torch.arange(len(zen))


# ---------------------------------------------------------------------------
# a6: PixInterp.get_interp / interp, rect grid             utils.py:742-861, 949-1116
# ---------------------------------------------------------------------------
_DEG = {'nearest': 0, 'linear': 1, 'quadratic': 2, 'cubic': 3}        # utils.py:780


def _stencil_start(t, n, N, wrap):
    """
    First node of the n-point stencil the reference selects for fractional grid index t:
    the n nearest nodes by |delta| (argsort, ties -> lower index), sorted ascending
    (utils.py:1003-1004).  Even n: window ends at ceil(t)+n/2-1 ... i.e. starts at
    ceil(t) - n/2; odd n: centred on the nearest node, ties toward the lower node.  A
    non-periodic axis slides the window back inside [0, N-n]; a periodic axis is extended by
    n nodes on both sides first (utils.py:1000), which is enough for any t in [0, N).
    """
    if n % 2 == 0:
        start = torch.ceil(t).to(torch.int64) - n // 2
    else:
        start = torch.ceil(t - 0.5).to(torch.int64) - n // 2
    if not wrap:
        start = start.clamp(0, N - n)
    return start


def _lagrange_weights(u, n):
    """weights of nodes 0..n-1 for the degree n-1 interpolating polynomial at u; (P, n)"""
    cols = []
    for j in range(n):
        w = torch.ones_like(u)
        for k in range(n):
            if k != j:
                w = w * (u - k) / (j - k)
        cols.append(w)
    return torch.stack(cols, dim=-1)


def rect_interp_weights(theta_grid, phi_grid, zen, az, interp_mode):
    """
    (inds (P, Nnn) int64, wgts (P, Nnn)) for bi-polynomial interpolation on a uniform
    (theta, phi) grid stored phi-fastest (flat index = iphi + Nphi * itheta).  The reference
    obtains the weights as Anew @ pinv(A^T A) A^T on the stencil (utils.py:1084-1116), which is
    the tensor-product Lagrange basis; stencil order is theta-slow, phi-fast (utils.py:1017).
    First token of a mixed mode ('linear,quadratic') is azimuth (utils.py:774-792).
    """
    if ',' in interp_mode:
        dx_, dy_ = [_DEG[s.strip()] for s in interp_mode.split(',')]
    else:
        dx_ = dy_ = _DEG[interp_mode]
    nx, ny = dx_ + 1, dy_ + 1
    Nphi, Nth = len(phi_grid), len(theta_grid)
    dphi = phi_grid[1] - phi_grid[0]
    dth = theta_grid[1] - theta_grid[0]
    tx = (az - phi_grid[0]) / dphi
    ty = (zen - theta_grid[0]) / dth
    sx = _stencil_start(tx, nx, Nphi, wrap=True)
    sy = _stencil_start(ty, ny, Nth, wrap=False)
    wx = _lagrange_weights(tx - sx.to(tx.dtype), nx)                 # xrel, utils.py:1007
    wy = _lagrange_weights(ty - sy.to(ty.dtype), ny)
    ix = (sx[:, None] + torch.arange(nx)) % Nphi                     # unwrap, utils.py:1011-1013
    iy = sy[:, None] + torch.arange(ny)
    inds = (ix[:, None, :] + Nphi * iy[:, :, None]).reshape(len(zen), -1)
    wgts = (wx[:, None, :] * wy[:, :, None]).reshape(len(zen), -1)
    return inds, wgts


def interp(m, inds, wgts):
    """out[..., p] = sum_k wgts[p, k] m[..., inds[p, k]].  utils.py:833-841"""
    nearest = m.index_select(-1, inds.reshape(-1)).reshape(m.shape[:-1] + inds.shape)
    return torch.einsum('...i,...i->...', nearest, wgts.to(nearest.dtype))


# ---------------------------------------------------------------------------
# HEALPix RING (parity unpinned: healpy absent)            utils.py:765-769 call site
# ---------------------------------------------------------------------------
def healpix_pix2ang(nside):
    """(colat, lon) [rad] of RING pixel centres (Gorski et al. 2005); float64 numpy"""
    npix = 12 * nside * nside
    ncap = 2 * nside * (nside - 1)
    p = np.arange(npix, dtype=np.int64)
    z = np.empty(npix)
    phi = np.empty(npix)
    # north cap
    c = p < ncap
    pc = p[c]
    i = np.floor((1 + np.sqrt(1 + 2 * pc.astype(np.float64))) / 2).astype(np.int64)
    j = pc + 1 - 2 * i * (i - 1)
    z[c] = 1.0 - i * i / (3.0 * nside * nside)
    phi[c] = (j - 0.5) * np.pi / (2.0 * i)
    # equatorial belt
    e = (p >= ncap) & (p < npix - ncap)
    pe = p[e] - ncap
    i = pe // (4 * nside) + nside
    j = pe % (4 * nside) + 1
    s = np.where((i + nside) % 2 == 1, 1.0, 0.5)
    z[e] = (2 * nside - i) * 2.0 / (3.0 * nside)
    phi[e] = (j - s) * np.pi / (2.0 * nside)
    # south cap
    sc = p >= npix - ncap
    ps = npix - p[sc]
    i = np.floor((1 + np.sqrt(2 * ps.astype(np.float64) - 1)) / 2).astype(np.int64)
    j = 4 * i + 1 - (ps - 2 * i * (i - 1))
    z[sc] = -1.0 + i * i / (3.0 * nside * nside)
    phi[sc] = (j - 0.5) * np.pi / (2.0 * i)
    return np.arccos(z), phi


# ---------------------------------------------------------------------------
# a9: sph_harm.gen_lm / gen_sph2pix (integer l, full sphere) / AlmModel.forward_alm
# ---------------------------------------------------------------------------
def gen_lm(lmax):
    """m-major (l, m) list, m >= 0.  sph_harm.py:14-40"""
    l, m = [], []
    for mm in range(0, lmax + 1):
        for ll in range(mm, lmax + 1):
            l.append(ll)
            m.append(mm)
    return np.array(l), np.array(m)


def sph_Ylm(theta, phi, l, m):
    """
    Orthonormal complex Y_lm(theta, phi) with Condon-Shortley phase, integer l >= m >= 0,
    (Ncoeff, Npix) complex128; theta colatitude, radians.  Same function as the reference's
    gen_sph2pix(method='sphere') (sph_harm.py:255-475: sqrt((2l+1)/4pi (l-m)!/(l+m)!)
    P_lm(cos theta) e^{i m phi}), evaluated here with the standard stable recurrences on the
    normalised functions instead of hypergeometric series.
    """
    theta = np.asarray(theta, dtype=np.float64)
    phi = np.asarray(phi, dtype=np.float64)
    x, sth = np.cos(theta), np.sin(theta)
    lmax, mmax = int(np.max(l)), int(np.max(m))
    N = {}                                   # normalised P~_lm
    pmm = np.full_like(x, math.sqrt(1.0 / (4 * math.pi)))
    for mm in range(0, mmax + 1):
        if mm > 0:
            pmm = -math.sqrt((2 * mm + 1) / (2.0 * mm)) * sth * pmm
        N[(mm, mm)] = pmm
        if mm + 1 <= lmax:
            N[(mm + 1, mm)] = math.sqrt(2 * mm + 3) * x * pmm
        for ll in range(mm + 2, lmax + 1):
            a = math.sqrt((4.0 * ll * ll - 1) / (ll * ll - mm * mm))
            b = math.sqrt(((ll - 1.0) ** 2 - mm * mm) / (4.0 * (ll - 1) ** 2 - 1))
            N[(ll, mm)] = a * (x * N[(ll - 1, mm)] - b * N[(ll - 2, mm)])
    Y = np.empty((len(l), len(theta)), dtype=np.complex128)
    for k, (ll, mm) in enumerate(zip(l, m)):
        Y[k] = N[(int(ll), int(mm))] * np.exp(1j * mm * phi)
    return Y


def alm_mult(m):
    """1 for m == 0, 2 for m > 0 (negative m folded in for a real field).  sph_harm.py:468-471"""
    return np.where(np.asarray(m) > 0, 2.0, 1.0)


def forward_alm(params, Ylm, mult=None, real_output=True):
    """
    (params * mult) @ Ylm, then .real.  params (..., Ncoeff) complex or (..., Ncoeff, 2) real
    view; Ylm (Ncoeff, Npix) or a (Theta (Ncoeff, Ntheta), Phi (Ncoeff, Nphi)) pair for a
    separable grid (output then theta-slow, phi-fast).  sph_harm.py:1342-1372
    """
    separable = isinstance(Ylm, (tuple, list))
    Yc = Ylm[1] if separable else Ylm
    if torch.is_complex(Yc) and not torch.is_complex(params):
        params = torch.view_as_complex(params)
    if mult is not None:
        params = params * mult
    if separable:
        Theta, Phi = Ylm
        t = torch.einsum('ct,...c->...tc', Theta.to(params.dtype), params)
        out = torch.einsum('...tc,cp->...tp', t, Phi.to(params.dtype))
        out = out.reshape(out.shape[:-2] + (Theta.shape[1] * Phi.shape[1],))
    else:
        out = torch.einsum('...i,ij->...j', params, Ylm.to(params.dtype))
    return out.real if real_output else out


# ---------------------------------------------------------------------------
# a12 / analytic responses (inputs to the path)
# ---------------------------------------------------------------------------
def point_powerlaw(params, freqs, f0, log=False):
    """amp (f/f0)^alpha; params (..., 2, Nsrc).  sky_model.py:353-357"""
    amp = params[..., 0:1, :]
    if log:
        amp = torch.exp(amp)
    return amp * (freqs[:, None] / f0) ** params[..., 1:2, :]


def airy_beam(zen, az, Dew, freqs, Dns=None, square=True):
    """[2 J1(x)/x]^(2|1), x = pi D nu sin(zen)/c, zen clipped at the horizon.  beam_model.py:1418-1482"""
    z = (zen * D2R).clone()
    z[z > math.pi / 2] = math.pi / 2
    if Dns is None:
        diameter = Dew
    else:
        diameter = Dns + torch.abs(torch.sin(az * D2R)) ** 2 * (Dew - Dns)
    x = (diameter * torch.sin(z) * math.pi * freqs.reshape(-1, 1) / C_LIGHT).clip(1e-10)
    b = 2.0 * torch.special.bessel_j1(x) / x
    return b ** 2 if square else b


def gauss_beam(zen, az, params, powerbeam=True):
    """exp(-0.5((l/sig_ew)^2 + (m/sig_ns)^2)); params (..., Nf, 2).  beam_model.py:886-896"""
    zr, ar = zen * D2R, az * D2R
    srad = torch.sin(zr).clone()
    srad[zr > math.pi / 2] = 1.0
    l = srad * torch.sin(ar)
    m = srad * torch.cos(ar)
    b = torch.exp(-0.5 * ((l / params[..., 0:1]) ** 2 + (m / params[..., 1:2]) ** 2))
    return b if powerbeam else torch.sqrt(b)


def pixel_response_forward(params, log=False, powerbeam=True, realbeam=True, beam0=None,
                           norm_pix=None):
    """beam map from params: .real -> exp | abs -> + beam0 -> / |pix|.  beam_model.py:750-793"""
    p = params
    if torch.is_complex(p) and (realbeam or powerbeam):
        p = p.real
    if log:
        p = torch.exp(p)
    elif powerbeam:
        p = torch.abs(p)
    if beam0 is not None:
        p = p + beam0
    if norm_pix is not None:
        p = p / p[..., norm_pix:norm_pix + 1].detach().abs()
    return p


def stokes_to_coherency(I, frac):
    """[[I+Q, U-iV],[U+iV, I-Q]] with Q = I fQ ...; I (Nf, P); frac (<=3, Nf, P).  sky_model.py:1244-1290"""
    Q = I * frac[0]
    U = I * frac[1] if len(frac) > 1 else torch.zeros_like(I)
    if len(frac) > 2:
        V = I * frac[2]
        B = torch.stack([torch.stack([I + Q, U - 1j * V]), torch.stack([U + 1j * V, I - Q])])
    else:
        B = torch.stack([torch.stack([I + Q, U]), torch.stack([U, I - Q])])
    return B


# ---------------------------------------------------------------------------
# a1: RIME.forward                                          rime_model.py:291-389
# ---------------------------------------------------------------------------
def rime_forward(sky, zenaz, beam_fn, blvecs, bl_models, freqs, powerbeam=True, fov=180.0,
                 sim2data_idx=None):
    """
    sky (Nv, Nv, Nf, Npix) flux density per pixel (already x px_area); zenaz (Nt, 2, Npix)
    [deg] per time; beam_fn(zen, az) -> (Npol, Nvec, Nmodel, Nf, P) evaluated on the
    FoV-cut angles (gen_beam, beam_model.py:221-259).  Loops times like rime_model.py:334-365
    and stacks on dim 3 (:368).  Returns vis (Npol, Npol|1, Nbl, Nt, Nf).
    """
    out = []
    for t in range(len(zenaz)):
        zen, az = zenaz[t, 0], zenaz[t, 1]
        cut = fov_cut(zen, fov)
        zc, ac = zen[cut], az[cut]
        beam = beam_fn(zc, ac)
        cut_sky = sky.index_select(-1, cut)                          # beam_model.py:1696
        psky = apply_beam(beam, cut_sky, bl_models, powerbeam)
        out.append(prod_and_sum(psky, blvecs, zc, ac, freqs, sim2data_idx))
    return torch.stack(out, dim=3)


def apply_icov_diag(res, icov=None):
    """optim.apply_icov with cov_axis=None (optim.py:1889-1894): conj(res) * res [* icov], elementwise"""
    out = res.conj() * res
    return out if icov is None else out * icov


def chisq(pred, data, icov=None):
    """LogProb.forward_chisq's value (optim.py:1019-1027): sum of apply_icov(pred - data), real part"""
    tot = torch.sum(apply_icov_diag(pred - data, icov))
    return tot.real if torch.is_complex(tot) else tot


def build_A(blvecs, zen, az, freqs, beam=None):
    """imaging matrix of one time step, (Nbl, Nf, P): conj(fringe) [* beam] (imaging.py:251-296)"""
    A = gen_fringe(blvecs, zen, az, freqs, conj=True)
    return A if beam is None else A * beam


def make_map(v, w, A):
    """dirty map Re sum_b A (v w), (..., Nf, P) (imaging.py:717-736)"""
    return torch.einsum('vfp,...vf->...fp', A, (v * w).to(A.dtype)).real


def compute_Am(A, m):
    """conj(A) @ m: the RIME forward of a map, (..., Nbl, Nf) (imaging.py:755-774)"""
    return torch.einsum('vfp,...fp->...vf', A.conj(), m.to(A.dtype))


def compute_Pm(A, w, m, D=None):
    """P m = D A^T w (conj(A) m), (..., Nf, P) (imaging.py:777-815)"""
    Pm = torch.einsum('vfp,...vf->...fp', A, w * compute_Am(A, m)).real
    return Pm if D is None else Pm * D


def compute_P(A, w, contract=None):
    """PSF matrix A^T w conj(A) of one time step: full (Nf, P, P), its diagonal or its row sums (imaging.py:818-861)"""
    if contract is None:
        return torch.einsum('vfp,vfq->fpq', A, w[..., None] * A.conj()).real
    if contract == 'diag':
        return (w[..., None] * A.abs().pow(2)).sum(0)
    return torch.einsum('vfp,vf,vfq->fp', A, w.to(A.dtype).expand(A.shape[:2]), A.conj()).real


def vismapper_make_map(blvecs, zenaz, freqs, vis, weights, beam_fn=None, fov=180.0, method='A2w', clip=1e-8,
                       contract='diag'):
    """
    VisMapper.make_map (imaging.py:360-466): per time step t the cut (beam: zen < fov/2 through gen_beam; no beam:
    zen <= fov/2, :268-283), A_t = conj(fringe) beam, dirty map and PSF contraction scattered into the full pixel
    axis, summed over t and divided by the normalisation sum -- 'w': sum_b w, 'Aw': sum_b w |A|, 'A2w': sum_b w Re(A^2)
    (:442-447: the real part of the SQUARE, not |A|^2 as compute_Pm / compute_P use at :619 / :694) -- clipped from below.
    vis (..., Nbl, Nt, Nf) complex, weights (Nbl, Nt, Nf), zenaz (Nt, 2, Npix) deg, beam_fn(zen, az) -> (Nf, P).
    Returns (maps (..., Nf, Npix), P, D).
    """
    Nt, Npix, Nf = zenaz.shape[0], zenaz.shape[-1], len(freqs)
    maps = torch.zeros(vis.shape[:-3] + (Nf, Npix), dtype=weights.dtype)
    Aw = torch.zeros(Nf, 1 if method == 'w' else Npix, dtype=weights.dtype)
    P = None if contract == 'none' else torch.zeros((Nf, Npix) + ((Npix,) if contract is None else ()), dtype=weights.dtype)
    for t in range(Nt):
        zen, az = zenaz[t]
        if beam_fn is not None:
            cut = fov_cut(zen, fov)
            beam = beam_fn(zen[cut], az[cut])
        else:
            cut, beam = torch.where(zen <= fov / 2)[0], None
        A = build_A(blvecs, zen[cut], az[cut], freqs, beam)
        w = weights[:, t]
        maps[..., cut] += make_map(vis[..., t, :], w, A)
        if P is not None:
            if contract is None:
                P[:, cut[:, None], cut[None, :]] += compute_P(A, w, None)
            else:
                P[:, cut] += compute_P(A, w, contract)
        if method == 'w':
            Aw += w.sum(0)[:, None]
        elif method == 'Aw':
            Aw[..., cut] += (w[:, :, None] * A.abs()).sum(0)
        else:
            Aw[..., cut] += (w[:, :, None] * A.pow(2).real).sum(0)
    D = 1 / Aw.clip(clip)
    maps = maps * D
    if P is not None:
        P = P * (D[:, :, None] if contract is None else D)
    return maps, P, D


def apply_cal(vis, gains, g1_idx, g2_idx, cal_2pol=False):
    """
    calibration._apply_cal for complex visibilities without undo / covariance
    (calibration.py:2412-2487): V'_pq = G_p V_pq G_q^dagger.  vis (Np, Np, Nbl, Nt, Nf); gains
    (Np, Np, Nant, Nt|1, Nf|1); 1-pol and 2-pol (cal_2pol: diagonal products only, off-diagonals of
    the result zero -- linalg.diag_matmul :116-149); 4-pol full 2x2 products.
    """
    g1 = gains.index_select(2, torch.as_tensor(g1_idx))
    g2 = gains.index_select(2, torch.as_tensor(g2_idx))
    if vis.shape[0] == 1:
        return g1 * g2.conj() * vis
    if cal_2pol:
        G = g1 * g2.conj()
        out = torch.zeros_like(vis)
        out[0, 0] = G[0, 0] * vis[0, 0]
        out[1, 1] = G[1, 1] * vis[1, 1]
        return out
    return torch.einsum('ab...,bc...,dc...->ad...', g1, vis, g2.conj())
