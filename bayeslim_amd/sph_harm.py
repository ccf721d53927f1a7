"""
Spherical-harmonic forward model with the reference's API (sph_harm.py): `gen_lm` (:14-40),
`gen_sph2pix` for integer degree on the full sphere (:255-475) and `AlmModel`
(:1244-1581) whose a_lm -> pixel product runs in the HIP kernel `rime_alm2pix_fwd/bwd`.

Out of scope (one-off host setup in the reference): cut-sky (cap / stripe)
non-integer-degree bases built from hypergeometric functions, spherical Fourier-Bessel models,
HDF5 Ylm files.
"""
import math

import numpy as np
import torch

from . import utils, ops
from .utils import _float, _cfloat, D2R


def gen_lm(lmax, real_field=True):
    """(2, Ncoeff) array of (l, m), m-major, m >= 0 for a real field (sph_harm.py:14-40)"""
    lm = [(l, m) for m in range(0 if real_field else -lmax, lmax + 1)
          for l in range(abs(m), lmax + 1)]
    return np.array(lm).T


def _norm_legendre(x, sth, lmax, mmax):
    """orthonormalised P~_lm(x) (incl. sqrt((2l+1)/4pi (l-m)!/(l+m)!) and the Condon-Shortley
    phase) for 0 <= m <= mmax, m <= l <= lmax, by upward recurrence in l; dict[(l, m)]"""
    out = {}
    pmm = np.full_like(x, math.sqrt(1.0 / (4 * math.pi)))
    for m in range(mmax + 1):
        if m > 0:
            pmm = -math.sqrt((2 * m + 1) / (2.0 * m)) * sth * pmm
        out[(m, m)] = pmm
        if m + 1 <= lmax:
            out[(m + 1, m)] = math.sqrt(2 * m + 3) * x * pmm
        for l in range(m + 2, lmax + 1):
            a = math.sqrt((4.0 * l * l - 1) / (l * l - m * m))
            b = math.sqrt(((l - 1.0) ** 2 - m * m) / (4.0 * (l - 1) ** 2 - 1))
            out[(l, m)] = a * (x * out[(l - 1, m)] - b * out[(l - 2, m)])
    return out


def gen_sph2pix(theta, phi, l, m, separable=False, method='sphere', device=None, real=False,
                m_phasor=False, **kwargs):
    """
    Y_lm(theta, phi) matrix for integer l on the full sphere (theta colatitude [rad], phi
    [rad]).  Returns (Ylm, norm, alm_mult) like the reference (sph_harm.py:255-475):
    Ylm (Ncoeff, Npix) complex (or real part if real=True), or (Theta (Ncoeff, Ntheta),
    Phi (Ncoeff, Nphi)) if separable; alm_mult = 2 for m > 0 when negative m are truncated.
    """
    if method != 'sphere':
        raise NotImplementedError("only method='sphere' (integer degree) is built natively; "
                                  "cut-sky bases are out of scope")
    l = np.atleast_1d(np.asarray(l))
    m = np.atleast_1d(np.asarray(m))
    assert np.allclose(l, np.round(l)) and np.allclose(m, np.round(m)), 'integer l, m only'
    theta = np.atleast_1d(np.asarray(utils.tensor2numpy(theta), dtype=np.float64))
    phi = np.atleast_1d(np.asarray(utils.tensor2numpy(phi), dtype=np.float64))
    li, mi = l.astype(int), m.astype(int)
    N = _norm_legendre(np.cos(theta), np.sin(theta), int(li.max()), int(np.abs(mi).max()))
    H = np.empty((len(l), len(theta)))
    for k, (ll, mm) in enumerate(zip(li, mi)):
        h = N[(ll, abs(mm))]
        H[k] = h if mm >= 0 else (-1) ** abs(mm) * h      # Y_{l,-m} = (-1)^m conj(Y_lm)
    Phi = np.exp(1j * mi[:, None] * phi[None, :])
    if m_phasor:
        Phi = Phi * np.exp(1j * phi)[None, :]
    dtype = _float() if real else _cfloat()
    if separable:
        Y = (torch.as_tensor(H, dtype=dtype, device=device),
             torch.as_tensor(Phi.real if real else Phi, dtype=dtype, device=device))
    else:
        full = H * Phi
        Y = torch.as_tensor(full.real if real else full, dtype=dtype, device=device)
    norm = torch.ones(len(l))
    alm_mult = torch.ones(len(l), dtype=_float())
    if not np.any(mi < 0) and not real:
        alm_mult[mi > 0] *= 2
    if m_phasor and not real:
        alm_mult[mi == 0] *= 2
    return Y, norm, alm_mult


def inflate_Ylm(Ylm):
    """(Theta, Phi) -> full (Ncoeff, Ntheta*Nphi), theta-slow / phi-fast"""
    if isinstance(Ylm, (tuple, list)):
        T, P = Ylm
        return (T[:, :, None] * P[:, None, :]).reshape(T.shape[0], -1)
    return Ylm


class AlmModel:
    """
    f(theta, phi) = sum_lm a_lm Y_lm: params (..., Ncoeff) [complex, or (..., Ncoeff, 2) real
    view] -> map (..., Npix).  Ylm matrices are cached per angle set (sph_harm.py:1244-1581).
    """
    def __init__(self, l, m, default_kw=None, real_output=False, LM=None):
        self.l, self.m = l, m
        self.device = None
        self.default_kw = {} if default_kw is None else default_kw
        self.real_output = real_output
        self.LM = LM
        self.clear_Ylm_cache()
        self.clear_multigrid()

    def __call__(self, params, **kwargs):
        return self.forward_alm(params, **kwargs)

    def clear_Ylm_cache(self):
        self.Ylm_cache = {}

    def __getstate__(self):
        # pickle / deepcopy: the per-object conversion caches are keyed on id() of the ORIGINAL's Ylm tensors (and would carry
        # a second copy of each matrix along); the copy converts again on first use.  Ylm_cache itself travels, as in the
        # reference (sph_harm.py:1244-1581 keeps it on the object that io.write_pkl pickles)
        state = dict(self.__dict__)
        for k in ('_Ylm_pack_cache', '_Ylm_cast_cache', '_inflated'):
            state.pop(k, None)
        state['_inflated_key'] = None
        return state

    def clear_multigrid(self):
        self.multigrid = None
        self._multigrid_idx = None

    def set_multigrid(self, keys, idx=None):
        self.multigrid = list(keys)
        self._multigrid_idx = idx

    def forward_alm(self, params, Ylm=None, alm_mult=None, ignoreLM=False):
        """(params * alm_mult) @ Ylm [-> .real]  (sph_harm.py:1289-1372)"""
        if self.LM is not None and not ignoreLM:
            params = self.LM(params)
        if Ylm is None and self.multigrid is not None:
            outs = []
            for h in self.multigrid:
                c = self.Ylm_cache[h]
                outs.append(self.forward_alm(params, Ylm=c['Ylm'], alm_mult=c['alm_mult'], ignoreLM=True))
            out = torch.cat(outs, dim=-1)
            if self._multigrid_idx is not None:
                out = out.index_select(-1, self._multigrid_idx)
            return out
        if Ylm is None:
            Ylm, alm_mult = self.Ylm, self.alm_mult
        separable = isinstance(Ylm, (list, tuple))
        Yc = Ylm[1] if separable else Ylm
        if torch.is_complex(Yc) and not torch.is_complex(params):
            params = utils.viewcomp(params)
        if alm_mult is not None:
            params = params * alm_mult.to(params.device)
        if separable:
            # separable grids are small (Ntheta + Nphi columns): inflate once per Ylm object
            key = id(Ylm[0])
            if getattr(self, '_inflated_key', None) != key:
                self._inflated = inflate_Ylm(Ylm).contiguous()
                self._inflated_key = key
            Ylm = self._inflated
        if not params.is_cuda:
            raise RuntimeError('AlmModel.forward_alm needs GPU tensors (no CPU path)')
        if torch.is_complex(Ylm) and torch.is_complex(params):
            Yk = self._cast_Ylm(Ylm, params.dtype)
            if self.real_output:
                return ops.alm2pix(params, Yk)
            # complex output: Im(a Y) = Re((-i a) Y), so the rows [a ; -i a] go through the kernels in ONE
            # pass over Ylm (2 R rows) and the two halves are the real and imaginary parts
            both = ops.alm2pix(torch.stack([params, params * (-1j)]), Yk)
            return torch.complex(both[0], both[1])
        if not torch.is_complex(Ylm) and not torch.is_complex(params):
            # real Ylm (gen_sph2pix(real=True)): sum_c a_c Y_c = Re sum_k (a_2k - i a_2k+1)(Y_2k + i Y_2k+1) -- the
            # same kernels on a pair-packed copy of Ylm (built once per Ylm object, same bytes)
            Yp = self._packed_real_Ylm(Ylm, params.dtype)
            a = params
            if a.shape[-1] % 2:
                a = torch.cat([a, a.new_zeros(a.shape[:-1] + (1,))], dim=-1)
            a = a.reshape(a.shape[:-1] + (a.shape[-1] // 2, 2))
            return ops.alm2pix(torch.complex(a[..., 0], -a[..., 1]), Yp)
        # mixed real / complex operands (not produced by the reference's own constructors): plain GEMM
        out = torch.einsum('...i,ij->...j', params, Ylm.to(params.dtype))
        return out.real if (self.real_output and torch.is_complex(out)) else out

    def _packed_real_Ylm(self, Ylm, dtype):
        """real (Ncoeff, Npix) Ylm as complex (ceil(Ncoeff / 2), Npix): rows 2k + i rows 2k+1, cached per Ylm object"""
        cache = self.__dict__.setdefault('_Ylm_pack_cache', {})
        ent = cache.get(id(Ylm))
        if ent is None or ent[0] is not Ylm or ent[1] != Ylm._version or ent[2].real.dtype != dtype:
            if len(cache) > 8:
                cache.clear()
            Y = Ylm.to(dtype)
            if Y.shape[0] % 2:
                Y = torch.cat([Y, Y.new_zeros((1, Y.shape[1]))], dim=0)
            ent = (Ylm, Ylm._version, torch.complex(Y[0::2], Y[1::2]).contiguous())
            cache[id(Ylm)] = ent
        return ent[2]

    def _cast_Ylm(self, Ylm, dtype):
        """Ylm in the parameters' dtype, converted once per Ylm object (a complex128 matrix used with
        complex64 parameters would otherwise be re-cast -- 3.3 GB at C3 -- on every forward)"""
        if Ylm.dtype == dtype:
            return Ylm
        cache = self.__dict__.setdefault('_Ylm_cast_cache', {})
        ent = cache.get(id(Ylm))
        if ent is None or ent[0] is not Ylm or ent[1] != Ylm._version or ent[2].dtype != dtype:
            if len(cache) > 8:
                cache.clear()
            ent = (Ylm, Ylm._version, Ylm.to(dtype))
            cache[id(Ylm)] = ent
        return ent[2]

    @staticmethod
    def setup_angs(theta, phi, separable):
        if separable:
            ph, th = np.meshgrid(utils.tensor2numpy(phi), utils.tensor2numpy(theta), copy=False)
            return th.ravel(), ph.ravel()
        return theta, phi

    def setup_Ylm(self, theta, phi, Ylm=None, alm_mult=None, separable=False, generate=False,
                  cache=True, h=None, **kwargs):
        """attach (and optionally generate / cache) the transform for these angles [deg]
        (sph_harm.py:1408-1494)"""
        self.theta, self.phi = theta, phi
        if separable:
            self.theta_grid, self.phi_grid = theta, phi
            self.theta, self.phi = self.setup_angs(theta, phi, separable)
        if Ylm is None and generate:
            kw = dict(self.default_kw)
            kw.update(kwargs)
            th, ph = (self.theta_grid, self.phi_grid) if separable else (self.theta, self.phi)
            Ylm, _, alm_mult = gen_sph2pix(utils.tensor2numpy(th) * D2R, utils.tensor2numpy(ph) * D2R,
                                           self.l, self.m, separable=separable, device=self.device, **kw)
        self.Ylm, self.alm_mult, self.separable = Ylm, alm_mult, separable
        if cache:
            angs = (self.theta_grid, self.phi_grid) if separable else (theta, phi)
            self.set_Ylm(Ylm, angs, alm_mult=alm_mult, h=h)

    def get_Ylm(self, theta, phi, separable=False, h=None):
        h = h if h is not None else utils.arr_hash(theta)
        if h in self.Ylm_cache:
            c = self.Ylm_cache[h]
            self.Ylm, self.alm_mult = c['Ylm'], c['alm_mult']
            self.theta, self.phi = c['angs']
        else:
            self.setup_Ylm(theta, phi, cache=True, h=h, separable=separable, generate=True)
        self.separable = separable
        return self.Ylm, self.alm_mult

    def set_Ylm(self, Ylm, angs, alm_mult=None, h=None):
        h = h if h is not None else utils.arr_hash(angs[0])
        self.Ylm_cache[h] = dict(Ylm=Ylm, angs=angs, separable=isinstance(Ylm, (tuple, list)),
                                 alm_mult=alm_mult)
        return h

    def push(self, device):
        if not isinstance(device, torch.dtype):
            self.device = device
        mv = lambda y: tuple(utils.push(t, device) for t in y) if isinstance(y, (tuple, list)) \
            else utils.push(y, device)
        for c in self.Ylm_cache.values():
            c['Ylm'] = mv(c['Ylm']) if c['Ylm'] is not None else None
            if c['alm_mult'] is not None:
                c['alm_mult'] = utils.push(c['alm_mult'], device)
        if getattr(self, 'Ylm', None) is not None:
            self.Ylm = mv(self.Ylm)
        if getattr(self, 'alm_mult', None) is not None:
            self.alm_mult = utils.push(self.alm_mult, device)
        self._inflated_key = None
        self.__dict__.pop('_Ylm_cast_cache', None)
        if self.LM is not None:
            self.LM.push(device)
